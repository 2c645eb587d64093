// Controller.hpp — the drop-in boundary.  Public members are those of the reference's
// include/Controller.hpp:16-47, name for name and argument for argument, so ProgramHandler and
// RealtimeImageProcessing.cpp compile against it unchanged.  Behind it there is no OpenCL: every call
// forwards to the C-ABI of libmi355_imgfilter.so (include/mi355_imgfilter.h), i.e. to hand-written HIP
// kernels on an MI355X.
//
//   reference member (include/Controller.hpp)                  here
//   GetPlatforms / GetDevices            :21-22   one pseudo-platform; one device per visible GPU
//   CreateContext / CreateCommandQueue   :27-28   context record; queue = one mi355_ctx (GPU + stream)
//   CreateProgram(ctx, dev, filename)    :29      filter family from the .cl FILE NAME (no file is read)
//   CreateKernel(program, name)          :30      "grayscale" | "gaussian_blur" | "sobel_edge_detection"
//   PerformCLImageGrayscaling            :37-39   mi355_gray_rgba8
//   PerformCLImageEdgeDetection          :41-43   mi355_sobel_rgba8
//   PerformCLGaussianBlur                :45-47   mi355_gauss_rgba8
#ifndef CONTROLLER_H
#define CONTROLLER_H

#include <cstring>
#include <fstream>
#include <sstream>
#include <utility>
#include <vector>

#define _USE_MATH_DEFINES
#include <cmath>

#include <InfoPlatform.hpp>
#include <Logger.hpp>

class Controller
{
public:
    Controller();

    void CheckError(cl_int err, const char* name);

    std::vector<cl_platform_id> GetPlatforms();
    std::vector<cl_device_id> GetDevices(cl_platform_id platform);

    cl_bool GetImageSupport();
    void SetImageSupport(cl_bool image_support);

    cl_context CreateContext(cl_platform_id platform, std::vector<cl_device_id> devices);
    cl_command_queue CreateCommandQueue(cl_context context, cl_device_id device);
    cl_program CreateProgram(cl_context context, cl_device_id device, const char* filename);
    cl_kernel CreateKernel(cl_program program, const char* kernel_name);

    void DisplayPlatformInformation(cl_platform_id platform);
    void Cleanup(cl_context context = 0, cl_command_queue commandQueue = 0, cl_program program = 0,
                 cl_kernel kernel = 0, cl_sampler sampler = 0, cl_mem* mem_objects = 0, int num_mem_objects = 0);

    void PerformCLImageGrayscaling(cl_context* context, cl_command_queue* command_queue, cl_kernel* kernel,
                                   std::vector<cl_ulong>* profiling_events, std::vector<unsigned char>* input_data,
                                   std::vector<unsigned char>* output_data, cl_int& width, cl_int& height,
                                   Logger& logger);

    void PerformCLImageEdgeDetection(cl_context* context, cl_command_queue* command_queue, cl_kernel* kernel,
                                     std::vector<cl_ulong>* profiling_events, std::vector<unsigned char>* input_data,
                                     std::vector<unsigned char>* output_data, cl_int& width, cl_int& height,
                                     Logger& logger);

    void PerformCLGaussianBlur(int& kernel_size, float& kernel_sigma, cl_context* context,
                               cl_command_queue* command_queue, cl_kernel* kernel,
                               std::vector<cl_ulong>* profiling_events, std::vector<unsigned char>* input_data,
                               std::vector<unsigned char>* output_data, cl_int& width, cl_int& height,
                               Logger& logger);

    // The weight generator the reference's CPU path borrows from its Controller
    // (src/GaussianBlur/include/Controller.hpp:28, called at src/GaussianBlur/GaussianBlur.cpp:230).
    std::vector<float> _GenerateGausianKernel(int kernel_size, float sigma);

private:
    cl_uint num_platforms, num_devices;
    cl_bool m_image_support;
};

#endif  // CONTROLLER_H
