// Logger.hpp — the logger type every hot-path signature of the reference takes by reference
// (include/Controller.hpp:37-47 there), rebuilt with the same public surface as include/Logger.hpp:12-48 of the
// reference: singleton access, a level filter for the terminal, an optional append-mode file sink, and the three
// timing reports the application prints.  Line format and report texts are those of RT/src/Logger.cpp:60-136
// (the application's console output is part of its observable behaviour).
#pragma once

#include <string>
#include <mutex>
#include <fstream>
// the unchanged application relies on these arriving with the logger
#include <iostream>
#include <sstream>
#include <iomanip>
#include <memory>

class Logger
{
public:
    enum class LogLevel { INFO, WARNING, ERROR };

    // --- access: one instance per process, not copyable ---------------------------------------------------
    static Logger& getInstance();
    Logger(const Logger&) = delete;
    Logger& operator=(const Logger&) = delete;

    // --- sinks ----------------------------------------------------------------------------------------------
    // Opens `file_name` for appending; throws std::runtime_error if that fails.  Lines go to the file only
    // while `save_to_file` is set.
    void setLogFile(const std::string& file_name, bool save_to_file);
    // The terminal shows the messages whose level EQUALS the selected one, and only when enabled.
    void setTerminalDisplay(bool print_on_terminal);
    void setLogLevel(LogLevel level);

    // --- output ---------------------------------------------------------------------------------------------
    void log(const std::string& message, LogLevel level);  // "[Y-M-D h:m:s][LEVEL] message"
    std::string getCurrentTime();

    void PrintSummary(double& opencl_kernel_execution_time, double& opencl_kernel_write_time,
                      double& opencl_kernel_read_time, double& opencl_execution_time,
                      double& opencl_kernel_operation_time, double& cpu_execution_time);
    void PrintRawKernelExecutionTime(double& opencl_kernel_execution_time, double& opencl_kernel_write_time,
                                     double& opencl_kernel_read_time, double& opencl_kernel_operation_time);
    void PrintEndToEndExecutionTime(std::string method, double total_execution_time_ms);

private:
    Logger();
    ~Logger();

    struct Sinks {
        std::ofstream file;
        bool to_file = false;
        bool to_terminal = false;
        LogLevel terminal_level = LogLevel::INFO;
    };
    Sinks m_sinks;
    std::mutex m_guard;  // serialises log() and setLogFile()
};
