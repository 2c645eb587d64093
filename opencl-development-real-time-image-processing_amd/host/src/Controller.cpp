// Controller.cpp — the reference's Controller API (include/Controller.hpp:16-47) re-hosted on the C-ABI
// of libmi355_imgfilter.so.  Behaviour kept from RT/src/Controller.cpp (line numbers refer to it):
//   * bootstrap failures print "Error: <name> (<code>)" on std::cerr and exit(EXIT_FAILURE)   (:5-11)
//   * CreateCommandQueue / CreateProgram return NULL on failure                               (:122-125,:138-148)
//   * run-time failures inside Perform* are logged at ERROR level and the call returns        (:461-463,:484-486)
//   * every Perform* call appends exactly six cl_ulong ns timestamps to profiling_events:
//     write-start, write-end, kernel-start, kernel-end, read-start, read-end                  (:66-74,:471,:489,:513)
//   * Grayscale / EdgeDetection ASSIGN a new vector to *output_data whose length is the length the
//     caller pre-set (w*h*4 resp. w*h: RT/src/ProgramHandler.cpp:292,301); Gaussian writes in place
//     into the pre-sized vector                                                               (:510,:605,:731)
// Dropped on purpose: per-call clCreateBuffer/clReleaseMemObject (:234-244,:515-516), the second
// redundant H2D (:240 + :460), the leaked weights cl_mem (:674) and the single-threaded float->uchar host
// pass (:76-85) — device buffers are pooled in the context and the kernels emit u8 directly.
#include "Controller.hpp"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <string>

#include "handles.hpp"

namespace {

std::string lower_basename(const char* path)
{
    std::string s(path ? path : "");
    const size_t slash = s.find_last_of("/\\");
    if (slash != std::string::npos)
        s = s.substr(slash + 1);
    std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    return s;
}

mi355_ctx* queue_ctx(cl_command_queue* q)
{
    return (q && *q) ? (*q)->ctx : nullptr;
}

void append_profile(std::vector<cl_ulong>* events, const uint64_t prof[6])
{
    if (!events)
        return;
    for (int i = 0; i < 6; i++)
        events->push_back((cl_ulong)prof[i]);
}

}  // namespace

Controller::Controller() : num_platforms{0}, num_devices{0}, m_image_support{CL_FALSE} {}

void Controller::CheckError(cl_int err, const char* name)
{
    if (err != CL_SUCCESS) {
        std::cerr << "Error: " << name << " (" << err << ")" << std::endl;
        exit(EXIT_FAILURE);
    }
}

std::vector<cl_platform_id> Controller::GetPlatforms()
{
    num_platforms = 1;
    std::cout << "Number of platforms: " << num_platforms << std::endl;
    std::vector<cl_platform_id> platforms{mi355_host::the_platform()};
    std::cout << "Found " << platforms.size() << " platforms" << std::endl;
    return platforms;
}

std::vector<cl_device_id> Controller::GetDevices(cl_platform_id platform)
{
    std::vector<cl_device_id> devices;
    if (platform != mi355_host::the_platform())
        CheckError(CL_INVALID_VALUE, "clGetDeviceIDs");
    num_devices = (cl_uint)mi355_host::device_count();
    // the reference's second clGetDeviceIDs call fails when no device exists (:55-56): no silent fallback
    CheckError(num_devices == 0 ? CL_DEVICE_NOT_FOUND : CL_SUCCESS, "clGetDeviceIDs");
    for (cl_uint i = 0; i < num_devices; i++)
        devices.push_back(mi355_host::device_handle((int)i));
    std::cout << "Found " << devices.size() << " devices" << std::endl;
    return devices;
}

cl_bool Controller::GetImageSupport() { return m_image_support; }

void Controller::SetImageSupport(cl_bool image_support) { m_image_support = image_support; }

cl_context Controller::CreateContext(cl_platform_id platform, std::vector<cl_device_id> devices)
{
    CheckError((platform == mi355_host::the_platform() && !devices.empty()) ? CL_SUCCESS : CL_INVALID_VALUE,
               "clCreateContext");
    cl_context context = new _cl_context{(int)devices.size()};
    std::cout << "Successfully created a context" << std::endl;
    return context;
}

cl_command_queue Controller::CreateCommandQueue(cl_context context, cl_device_id device)
{
    mi355_ctx* ctx = nullptr;
    if (!context || !device || mi355_ctx_create(device->index, &ctx) != MI355_OK) {
        std::cerr << "Failed to create CommandQueue" << std::endl;
        return NULL;
    }
    cl_command_queue queue = new _cl_command_queue{ctx, device->index};
    std::cout << "Successfully created CommandQueue" << std::endl;
    return queue;
}

cl_program Controller::CreateProgram(cl_context context, cl_device_id device, const char* filename)
{
    // The reference compiles the OpenCL-C text of `filename` at run time (:131-179).  Here the kernels
    // are pre-compiled gfx950 code objects; the file NAME only selects the filter family, the file
    // itself is not opened (KERNELS[method][0|1] of the app: "grayscale_base.cl", "edge_images.cl", ...).
    (void)device;
    if (!context || !filename) {
        std::cerr << "Failed to create program objects from source" << std::endl;
        return NULL;
    }
    const std::string name = lower_basename(filename);
    int family = -1;
    if (name.find("gray") != std::string::npos)
        family = MI355_FAMILY_GRAY;
    else if (name.find("gauss") != std::string::npos)
        family = MI355_FAMILY_GAUSS;
    else if (name.find("edge") != std::string::npos || name.find("sobel") != std::string::npos)
        family = MI355_FAMILY_EDGE;
    if (family < 0) {
        std::cerr << "Failed to open file for reading: " << filename << std::endl;
        std::cerr << "Error: no filter family matches this kernel file name." << std::endl;
        return NULL;
    }
    std::cout << "Successfully created a program" << std::endl;
    return new _cl_program{family};
}

cl_kernel Controller::CreateKernel(cl_program program, const char* kernel_name)
{
    int family = -1;
    const std::string name(kernel_name ? kernel_name : "");
    if (name == "grayscale")
        family = MI355_FAMILY_GRAY;
    else if (name == "sobel_edge_detection")
        family = MI355_FAMILY_EDGE;
    else if (name == "gaussian_blur")
        family = MI355_FAMILY_GAUSS;
    // a kernel name that the program does not contain is CL_INVALID_KERNEL_NAME in the reference (:186-187)
    CheckError((program && family >= 0 && family == program->family) ? CL_SUCCESS : CL_INVALID_KERNEL_NAME,
               "clCreateKernel");
    std::cout << "Successfully created the " << kernel_name << " kernel" << std::endl;
    return new _cl_kernel{family};
}

void Controller::DisplayPlatformInformation(cl_platform_id platform)
{
    InfoPlatform platform_handler(platform);
    platform_handler.Display();
}

void Controller::Cleanup(cl_context context, cl_command_queue commandQueue, cl_program program, cl_kernel kernel,
                         cl_sampler sampler, cl_mem* mem_objects, int num_mem_objects)
{
    std::cout << "Performing cleanup" << std::endl;
    for (int i = 0; i < num_mem_objects; i++)
        if (mem_objects[i] != 0)
            clReleaseMemObject(mem_objects[i]);
    if (commandQueue != 0)
        clReleaseCommandQueue(commandQueue);
    if (kernel != 0)
        clReleaseKernel(kernel);
    if (program != 0)
        clReleaseProgram(program);
    if (context != 0)
        clReleaseContext(context);
    if (sampler != 0)
        clReleaseSampler(sampler);
    std::cout << "Succesfully cleaned environment" << std::endl;
}

std::vector<float> Controller::_GenerateGausianKernel(int kernel_size, float sigma)
{
    std::vector<float> table((size_t)std::max(kernel_size, 0) * std::max(kernel_size, 0));
    // the dispatcher of RT/src/Controller.cpp:405-427: image-mode generator when image support is on
    const int rc = (m_image_support == CL_TRUE) ? mi355_gauss_weights_image2d(kernel_size, sigma, table.data())
                                                : mi355_gauss_weights(kernel_size, sigma, table.data());
    if (rc != MI355_OK) {
        std::cerr << "Failed to create Gaussian Kernel" << std::endl;
        exit(1);
    }
    return table;
}

void Controller::PerformCLImageGrayscaling(cl_context* context, cl_command_queue* command_queue, cl_kernel* kernel,
                                           std::vector<cl_ulong>* profiling_events,
                                           std::vector<unsigned char>* input_data,
                                           std::vector<unsigned char>* output_data, cl_int& width, cl_int& height,
                                           Logger& logger)
{
    (void)context;
    mi355_ctx* ctx = queue_ctx(command_queue);
    const size_t npx = (size_t)std::max(width, 0) * std::max(height, 0);
    if (!ctx || !input_data || !output_data || npx == 0 || input_data->size() < npx * 4) {
        logger.log("Failed to write cl_mem (buffer) to kernel", Logger::LogLevel::ERROR);
        return;
    }
    if (!kernel || !*kernel || (*kernel)->family != MI355_FAMILY_GRAY)
        logger.log("Failed to set kernel arguments", Logger::LogLevel::ERROR);
    uint64_t prof[6] = {};
    if (m_image_support == CL_TRUE) {
        // image2d_t mode (:437-510 with m_image_support == CL_TRUE): the kernel writes an R/FLOAT image, the host reads
        // w*h floats into a zeroed vector of output_data->size() floats and truncates f * 255 — i.e. w*h gray bytes
        // followed by zeros; the caller views it as CV_8UC1 (RT/RealtimeImageProcessing.cpp:112)
        std::vector<unsigned char> gray(npx);
        const int rci = mi355_image2d_rgba8(ctx, MI355_FILTER_GRAY, input_data->data(), gray.data(), width, height, 0, 0.0f,
                                            prof);
        if (rci != MI355_OK) {
            logger.log(std::string("Failed when executing kernel: ") + mi355_strerror(rci), Logger::LogLevel::ERROR);
            return;
        }
        std::vector<unsigned char> result(output_data->size(), 0);
        std::copy_n(gray.begin(), std::min(result.size(), gray.size()), result.begin());
        *output_data = std::move(result);
        append_profile(profiling_events, prof);
        return;
    }
    std::vector<unsigned char> rgba(npx * 4);
    const int rc = mi355_gray_rgba8(ctx, input_data->data(), rgba.data(), width, height, prof);
    if (rc != MI355_OK) {
        logger.log(std::string("Failed when executing kernel: ") + mi355_strerror(rc), Logger::LogLevel::ERROR);
        return;
    }
    // buffer-mode shape (:442,:510): as many bytes as the caller pre-sized, (g,g,g,255) per pixel
    std::vector<unsigned char> result(output_data->size(), 0);
    std::copy_n(rgba.begin(), std::min(result.size(), rgba.size()), result.begin());
    *output_data = std::move(result);
    append_profile(profiling_events, prof);
}

void Controller::PerformCLImageEdgeDetection(cl_context* context, cl_command_queue* command_queue, cl_kernel* kernel,
                                             std::vector<cl_ulong>* profiling_events,
                                             std::vector<unsigned char>* input_data,
                                             std::vector<unsigned char>* output_data, cl_int& width, cl_int& height,
                                             Logger& logger)
{
    (void)context;
    mi355_ctx* ctx = queue_ctx(command_queue);
    const size_t npx = (size_t)std::max(width, 0) * std::max(height, 0);
    if (!ctx || !input_data || !output_data || npx == 0 || input_data->size() < npx * 4) {
        logger.log("Failed to write cl_mem (buffer) to kernel", Logger::LogLevel::ERROR);
        return;
    }
    if (!kernel || !*kernel || (*kernel)->family != MI355_FAMILY_EDGE)
        logger.log("Failed to set kernel arguments", Logger::LogLevel::ERROR);
    uint64_t prof[6] = {};
    std::vector<unsigned char> edges(npx);
    // image2d_t mode: red channel only, interior only, truncation (RT/kernel/edge_images.cl:3-47)
    const int rc = (m_image_support == CL_TRUE)
                       ? mi355_image2d_rgba8(ctx, MI355_FILTER_SOBEL, input_data->data(), edges.data(), width, height, 0,
                                             0.0f, prof)
                       : mi355_sobel_rgba8(ctx, input_data->data(), edges.data(), width, height, prof);
    if (rc != MI355_OK) {
        logger.log(std::string("Failed when executing kernel: ") + mi355_strerror(rc), Logger::LogLevel::ERROR);
        return;
    }
    std::vector<unsigned char> result(output_data->size(), 0);
    std::copy_n(edges.begin(), std::min(result.size(), edges.size()), result.begin());
    *output_data = std::move(result);
    append_profile(profiling_events, prof);
}

void Controller::PerformCLGaussianBlur(int& kernel_size, float& kernel_sigma, cl_context* context,
                                       cl_command_queue* command_queue, cl_kernel* kernel,
                                       std::vector<cl_ulong>* profiling_events, std::vector<unsigned char>* input_data,
                                       std::vector<unsigned char>* output_data, cl_int& width, cl_int& height,
                                       Logger& logger)
{
    (void)context;
    mi355_ctx* ctx = queue_ctx(command_queue);
    const size_t npx = (size_t)std::max(width, 0) * std::max(height, 0);
    if (!ctx || !input_data || !output_data || npx == 0 || input_data->size() < npx * 4 ||
        output_data->size() < npx * 4) {
        logger.log("Failed to write cl_mem (buffer) to kernel", Logger::LogLevel::ERROR);
        return;
    }
    if (!kernel || !*kernel || (*kernel)->family != MI355_FAMILY_GAUSS)
        logger.log("Failed to set kernel arguments", Logger::LogLevel::ERROR);
    uint64_t prof[6] = {};
    // in place into the caller's pre-sized vector (:731); image2d_t mode: border-colour taps, the image-mode table,
    // no renormalisation (RT/kernel/gaussian_images.cl:1-36, Controller.cpp:374-403,663)
    const int rc = (m_image_support == CL_TRUE)
                       ? mi355_image2d_rgba8(ctx, MI355_FILTER_GAUSS, input_data->data(), output_data->data(), width, height,
                                             kernel_size, kernel_sigma, prof)
                       : mi355_gauss_rgba8(ctx, input_data->data(), output_data->data(), width, height, kernel_size,
                                           kernel_sigma, prof);
    if (rc != MI355_OK) {
        logger.log(std::string("Failed when executing kernel: ") + mi355_strerror(rc), Logger::LogLevel::ERROR);
        return;
    }
    append_profile(profiling_events, prof);
}
