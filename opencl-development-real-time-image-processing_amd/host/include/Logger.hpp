// Logger.hpp — API-compatible with the reference's include/Logger.hpp:12-48: a Logger& is a parameter
// of every hot-path signature (include/Controller.hpp:37-47), so the type, its LogLevel enum and its
// public methods keep their names and meaning.  Mutex-guarded singleton, optional file + terminal sinks,
// "[time][LEVEL] message" lines, terminal output only for the selected level (RT/src/Logger.cpp:60-78).
#ifndef LOGGER_HPP
#define LOGGER_HPP

#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>

class Logger
{
public:
    enum class LogLevel { INFO, WARNING, ERROR };

    Logger(const Logger&) = delete;
    Logger& operator=(const Logger&) = delete;

    std::string getCurrentTime();

    static Logger& getInstance();
    void setLogLevel(LogLevel level);
    void setLogFile(const std::string& file_name, bool save_to_file);  // throws std::runtime_error
    void setTerminalDisplay(bool print_on_terminal);
    void log(const std::string& message, LogLevel level);

    void PrintEndToEndExecutionTime(std::string method, double total_execution_time_ms);
    void PrintRawKernelExecutionTime(double& opencl_kernel_execution_time, double& opencl_kernel_write_time,
                                     double& opencl_kernel_read_time, double& opencl_kernel_operation_time);
    void PrintSummary(double& opencl_kernel_execution_time, double& opencl_kernel_write_time,
                      double& opencl_kernel_read_time, double& opencl_execution_time,
                      double& opencl_kernel_operation_time, double& cpu_execution_time);

private:
    std::ofstream m_log_file;
    std::mutex m_mutex;
    bool m_print_terminal = false;
    bool m_save_to_file = false;
    LogLevel m_set_level = LogLevel::INFO;

    Logger();
    ~Logger();

    std::string _printLogLevel(LogLevel level);
};

#endif  // LOGGER_HPP
