// ProgramHandler.hpp — the class the reference application drives (include/ProgramHandler.hpp:6-45 there), with
// the same public members so that RealtimeImageProcessing.cpp compiles against it unchanged: it holds the run
// flags, picks the device, and turns a method string ("GRAYSCALE" | "EDGE" | "GAUSSIAN") into the matching
// Controller::PerformCL* call.  Private state is this build's own.
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
    // Flags as the application passes them (RT/RealtimeImageProcessing.cpp:435 in the reference); the Gaussian
    // defaults are the reference's (k = 17, sigma = 6).
    ProgramHandler(int number_of_iterations, bool log_events, bool display_images, bool display_terminal_results,
                   bool bypass_image_support, int gaussian_kernel_size = 17, float gaussian_sigma = 6.0f);

    // --- set-up -------------------------------------------------------------------------------------------
    void SetDeviceProperties(int platform_index, int device_index);
    // kernel_index is the filter family ("GRAYSCALE" | "EDGE" | "GAUSSIAN"); `kernels` = {image, buffer} source
    // file names, kept as labels only — this build has no kernel sources to compile.
    void AddKernels(std::vector<std::string> kernels, std::string kernel_index);
    void InitLogger(Logger& logger, Logger::LogLevel level, bool save_to_file);
    // Platform / device pick, context, queue, "program" and "kernel" handles for `method`.
    void InitOpenCL(Controller& controller, cl_context* context, cl_command_queue* command_queue,
                    cl_program* program, cl_kernel* kernel, std::string method, Logger& logger);

    // --- the hot path -------------------------------------------------------------------------------------
    // One frame already in memory (the real-time loop).  Returns the filter output as the reference does.
    std::vector<unsigned char> PerformOpenCL(Controller& controller, const cv::Mat& input_frame, cl_context* context,
                                             cl_command_queue* command_queue, cl_kernel* kernel, cl_int& width,
                                             cl_int& height, Logger& logger, std::string method);
    // An image file, NUMBER_OF_ITERATIONS times, with the averaged profiling figures (the benchmark loop).
    std::vector<unsigned char> PerformOpenCL(Controller& controller, std::string image_path, cl_context* context,
                                             cl_command_queue* command_queue, cl_kernel* kernel,
                                             double& avg_opencl_execution_time, double& avg_opencl_kernel_write_time,
                                             double& avg_opencl_kernel_execution_time,
                                             double& avg_opencl_kernel_read_time, double& avg_opencl_kernel_operation,
                                             cl_int& width, cl_int& height, Logger& logger, std::string method);

    // --- MI355X extension: a batch sharded over several GPUs -------------------------------------------------
    // Nothing in the reference to match (one queue on one device, RT/src/ProgramHandler.cpp:108).  `rgba_frames`
    // holds nframes equally sized RGBA frames back to back; they are cut into contiguous ranges, one per entry of
    // `devices` (HIP ordinals; empty = every visible GPU; an ordinal may repeat), and every GPU streams its range
    // on its own host thread and stream (mi355_group_filter_batched).  method: as PerformOpenCL, plus "PIPELINE"
    // (gray -> Gaussian -> Sobel fused).  Output: the frames' results back to back, in input order.  Errors end
    // the process the way Controller::CheckError does.
    std::vector<unsigned char> PerformOpenCLBatch(const std::vector<unsigned char>& rgba_frames, int nframes,
                                                  cl_int width, cl_int height, Logger& logger, std::string method,
                                                  std::vector<int> devices = {});
    ~ProgramHandler();

private:
    struct Options {
        int iterations = 1;
        bool log_events = false, display_images = false, display_terminal_results = false;
        bool bypass_image_support = true;
        int platform_index = 0, device_index = 0;
        int gauss_k = 17;
        float gauss_sigma = 6.0f;
    };
    Options m_opt;
    std::vector<std::string> m_methods;                              // the three family names, in the reference's order
    std::map<std::string, std::vector<std::string>> m_kernel_files;  // family -> {image kernel, buffer kernel}

    struct mi355_group* m_group = nullptr;                           // PerformOpenCLBatch: created on first use
    std::vector<int> m_group_devices;

    void LoadFrame(std::string image_path, std::vector<unsigned char>* input_data, cl_int* width, cl_int* height,
                   Logger& logger);
};

#endif  // PROGRAMHANDLER_H
