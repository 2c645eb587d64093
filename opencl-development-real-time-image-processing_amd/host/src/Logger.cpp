// Logger.cpp — behaviour of the reference logger (RT/src/Logger.cpp:1-136): lines are
// "[Y-M-D h:m:s][LEVEL] message"; the terminal shows only messages of the selected level; the file,
// when enabled, gets everything; setLogFile throws std::runtime_error when the file cannot be opened.
#include "Logger.hpp"

#include <ctime>
#include <stdexcept>

namespace {

const char* level_name(Logger::LogLevel level)
{
    switch (level) {
    case Logger::LogLevel::INFO: return "INFO";
    case Logger::LogLevel::WARNING: return "WARNING";
    case Logger::LogLevel::ERROR: return "ERROR";
    }
    return "UNKNOWN";
}

}  // namespace

Logger& Logger::getInstance()
{
    static Logger instance;
    return instance;
}

Logger::Logger() = default;

Logger::~Logger()
{
    if (m_sinks.file.is_open())
        m_sinks.file.close();
}

std::string Logger::getCurrentTime()
{
    const std::time_t now = std::time(nullptr);
    std::tm tm_buf;
    localtime_r(&now, &tm_buf);
    std::ostringstream oss;
    oss << (1900 + tm_buf.tm_year) << "-" << (1 + tm_buf.tm_mon) << "-" << tm_buf.tm_mday << " " << tm_buf.tm_hour
        << ":" << tm_buf.tm_min << ":" << tm_buf.tm_sec;
    return oss.str();
}


void Logger::setLogFile(const std::string& file_name, bool save_to_file)
{
    std::lock_guard<std::mutex> lock(m_guard);
    m_sinks.to_file = save_to_file;
    if (m_sinks.file.is_open())
        m_sinks.file.close();
    m_sinks.file.open(file_name, std::ios::out | std::ios::app);
    if (!m_sinks.file)
        throw std::runtime_error("Failed to open log file: " + file_name);
}

void Logger::setTerminalDisplay(bool print_on_terminal) { m_sinks.to_terminal = print_on_terminal; }

void Logger::setLogLevel(LogLevel level) { m_sinks.terminal_level = level; }

void Logger::log(const std::string& message, LogLevel level)
{
    std::lock_guard<std::mutex> lock(m_guard);
    const std::string line = "[" + getCurrentTime() + "][" + level_name(level) + "] " + message;
    if (m_sinks.terminal_level == level && m_sinks.to_terminal)
        std::cout << line << std::endl;
    if (m_sinks.to_file && m_sinks.file.is_open())
        m_sinks.file << line << std::endl;
}

void Logger::PrintEndToEndExecutionTime(std::string method, double total_execution_time_ms)
{
    log("-------------------- START OF " + method + " EXECUTION TIME (end-to-end) DETAILS --------------------",
        LogLevel::INFO);
    std::ostringstream oss;
    oss << std::fixed << std::setprecision(3) << "Total execution time (end-to-end): " << total_execution_time_ms
        << " ms";
    log(oss.str(), LogLevel::INFO);
    log("-------------------- END OF " + method + " EXECUTION TIME (end-to-end) DETAILS --------------------",
        LogLevel::INFO);
}

void Logger::PrintRawKernelExecutionTime(double& opencl_kernel_execution_time, double& opencl_kernel_write_time,
                                         double& opencl_kernel_read_time, double& opencl_kernel_operation_time)
{
    log("-------------------- START OF KERNEL EXEUCTION DETAILS --------------------", LogLevel::INFO);
    const std::pair<const char*, double> rows[] = {{"Kernel write time: ", opencl_kernel_write_time},
                                                   {"Kernel execution time: ", opencl_kernel_execution_time},
                                                   {"Kernel read time: ", opencl_kernel_read_time},
                                                   {"Kernel complete operation time: ", opencl_kernel_operation_time}};
    for (const auto& r : rows) {
        std::ostringstream oss;
        oss << std::fixed << std::setprecision(5) << r.first << r.second << " ms";
        log(oss.str(), LogLevel::INFO);
    }
    log("-------------------- END OF KERNEL EXEUCTION DETAILS --------------------", LogLevel::INFO);
}

void Logger::PrintSummary(double& opencl_kernel_execution_time, double& opencl_kernel_write_time,
                          double& opencl_kernel_read_time, double& opencl_execution_time,
                          double& opencl_kernel_operation_time, double& cpu_execution_time)
{
    if (m_sinks.to_terminal)
        std::cout << "\n **************************************** START OF OpenCL SUMMARY "
                     "**************************************** "
                  << std::endl;
    PrintEndToEndExecutionTime("OpenCL", opencl_execution_time);
    PrintRawKernelExecutionTime(opencl_kernel_execution_time, opencl_kernel_write_time, opencl_kernel_read_time,
                                opencl_kernel_operation_time);
    if (m_sinks.to_terminal) {
        std::cout << " **************************************** END OF OpenCL SUMMARY "
                     "**************************************** "
                  << std::endl;
        std::cout << "\n **************************************** START OF CPU SUMMARY "
                     "**************************************** "
                  << std::endl;
    }
    PrintEndToEndExecutionTime("CPU", cpu_execution_time);
    if (m_sinks.to_terminal)
        std::cout << "\n **************************************** END OF CPU SUMMARY "
                     "**************************************** "
                  << std::endl;
}
