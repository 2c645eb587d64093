// ProgramHandler.hpp — same public surface as the reference's include/ProgramHandler.hpp:6-45: flag
// holder, InitOpenCL (device pick + method -> kernel name + kernel-file pick), and the two PerformOpenCL
// overloads that dispatch to Controller by method string ("GRAYSCALE" | "EDGE" | "GAUSSIAN").
#ifndef PROGRAMHANDLER_H
#define PROGRAMHANDLER_H

#include <Controller.hpp>
#include <Logger.hpp>

#include <map>
#include <string>
#include <vector>

#if __has_include(<opencv2/opencv.hpp>)
#include <opencv2/opencv.hpp>
#else
#include <cv_min.hpp>
#endif

class ProgramHandler
{
public:
    ProgramHandler(int number_of_iterations, bool log_events, bool display_images, bool display_terminal_results,
                   bool bypass_image_support, int gaussian_kernel_size = 17, float gaussian_sigma = 6.0f);

    void InitLogger(Logger& logger, Logger::LogLevel level, bool save_to_file);
    void InitOpenCL(Controller& controller, cl_context* context, cl_command_queue* command_queue,
                    cl_program* program, cl_kernel* kernel, std::string method, Logger& logger);

    void AddKernels(std::vector<std::string> kernels, std::string kernel_index);
    void SetDeviceProperties(int platform_index, int device_index);

    std::vector<unsigned char> PerformOpenCL(Controller& controller, std::string image_path, cl_context* context,
                                             cl_command_queue* command_queue, cl_kernel* kernel,
                                             double& avg_opencl_execution_time, double& avg_opencl_kernel_write_time,
                                             double& avg_opencl_kernel_execution_time,
                                             double& avg_opencl_kernel_read_time, double& avg_opencl_kernel_operation,
                                             cl_int& width, cl_int& height, Logger& logger, std::string method);

    std::vector<unsigned char> PerformOpenCL(Controller& controller, const cv::Mat& input_frame, cl_context* context,
                                             cl_command_queue* command_queue, cl_kernel* kernel, cl_int& width,
                                             cl_int& height, Logger& logger, std::string method);

private:
    bool LOG_EVENTS;
    bool DISPLAY_IMAGES;
    bool DISPLAY_TERMINAL_RESULTS;
    bool BYPASS_IMAGE_SUPPORT;

    int NUMBER_OF_ITERATIONS;
    int PLATFORM_INDEX;
    int DEVICE_INDEX;

    int GAUSSIAN_KERNEL_SIZE;
    float GAUSSIAN_SIGMA;

    std::vector<std::string> METHOD;
    std::map<std::string, std::vector<std::string>> KERNELS;

    void GetImageOpenCL(std::string image_path, std::vector<unsigned char>* input_data, cl_int* width,
                        cl_int* height, Logger& logger);
};

#endif  // PROGRAMHANDLER_H
