// ProgramHandler.cpp — the reference's ProgramHandler (RT/src/ProgramHandler.cpp:1-329) over the new
// Controller: flag holder, InitOpenCL (method -> kernel name, image-support probe / bypass -> kernel file),
// the N-iteration timed loop of PerformOpenCL(image_path, ...) and the one-frame PerformOpenCL(cv::Mat, ...).
// Differences, all deliberate:
//   * the timed loop reads the profiling events of the CURRENT iteration; the reference indexes [0..5] of a
//     vector it never clears (:219-221), so its published kernel/write/read averages repeat iteration 0;
//   * without OpenCV (this container) image files are read as binary PPM (P6) instead of cv::imread.
#include "ProgramHandler.hpp"

#include <cassert>
#include <chrono>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "../../../include/mi355_imgfilter.h"

ProgramHandler::ProgramHandler(int number_of_iterations, bool log_events, bool display_images,
                               bool display_terminal_results, bool bypass_image_support, int gaussian_kernel_size,
                               float gaussian_sigma)
{
    m_opt.iterations = number_of_iterations;
    m_opt.log_events = log_events;
    m_opt.display_images = display_images;
    m_opt.display_terminal_results = display_terminal_results;
    m_opt.bypass_image_support = bypass_image_support;
    m_opt.gauss_k = gaussian_kernel_size;
    m_opt.gauss_sigma = gaussian_sigma;
    m_methods = {"GRAYSCALE", "GAUSSIAN"};
}

void ProgramHandler::InitLogger(Logger& logger, Logger::LogLevel level, bool save_to_file)
{
    try {
        logger.setLogFile("RealtimeImageProcessing.log", save_to_file);
        logger.setLogLevel(level);
        logger.setTerminalDisplay(m_opt.display_terminal_results);
        logger.log("Initialised logger", Logger::LogLevel::INFO);
    } catch (const std::exception& e) {
        std::cerr << "Error setting log file: " << e.what() << std::endl;
    }
}

void ProgramHandler::AddKernels(std::vector<std::string> kernels, std::string kernel_index)
{
    m_kernel_files.insert({kernel_index, kernels});
}

void ProgramHandler::SetDeviceProperties(int platform_index, int device_index)
{
    m_opt.platform_index = platform_index;
    m_opt.device_index = device_index;
}

void ProgramHandler::InitOpenCL(Controller& controller, cl_context* context, cl_command_queue* command_queue,
                                cl_program* program, cl_kernel* kernel, std::string method, Logger& logger)
{
    auto platforms = controller.GetPlatforms();
    if (m_opt.platform_index < 0 || m_opt.platform_index >= (int)platforms.size())
        controller.CheckError(CL_INVALID_VALUE, "clGetPlatformIDs");
    auto devices = controller.GetDevices(platforms[m_opt.platform_index]);
    if (m_opt.device_index < 0 || m_opt.device_index >= (int)devices.size())
        controller.CheckError(CL_INVALID_DEVICE, "clGetDeviceIDs");

    if (m_opt.display_terminal_results) {
        for (auto&& platform : platforms)
            controller.DisplayPlatformInformation(platform);
        std::ostringstream oss;
        oss << "\nApplication will use:\nPLATFORM INDEX:\t" << m_opt.platform_index << "\nDEVICE INDEX:\t" << m_opt.device_index
            << "\n"
            << std::endl;
        logger.log(oss.str(), Logger::LogLevel::INFO);
        oss.str("");
        char device_name[256] = {};
        clGetDeviceInfo(devices[m_opt.device_index], CL_DEVICE_NAME, sizeof(device_name), device_name, NULL);
        oss << "Device name: " << device_name << std::endl;
        logger.log(oss.str(), Logger::LogLevel::INFO);
    }

    std::string kernel_name;
    if (method == "GRAYSCALE") {
        kernel_name = "grayscale";
    } else if (method == "EDGE") {
        kernel_name = "sobel_edge_detection";
    } else if (method == "GAUSSIAN") {
        kernel_name = "gaussian_blur";
    } else {
        std::cerr << "Unrecognised method" << std::endl;
        exit(1);
    }

    // m_kernel_files[method] = {image2d file, buffer file}; without an AddKernels entry fall back to the app's
    // own naming scheme so the family is still recognisable
    std::vector<std::string> files = m_kernel_files.count(method) ? m_kernel_files[method] : std::vector<std::string>{};
    if (files.size() < 2) {
        const std::string stem = method == "GRAYSCALE" ? "grayscale" : (method == "EDGE" ? "edge" : "gaussian");
        files = {stem + "_images.cl", stem + "_base.cl"};
    }
    std::string kernel_file;
    cl_bool image_support = CL_FALSE;
    if (!m_opt.bypass_image_support) {
        clGetDeviceInfo(devices[m_opt.device_index], CL_DEVICE_IMAGE_SUPPORT, sizeof(cl_bool), &image_support, nullptr);
        if (image_support == CL_FALSE) {
            kernel_file = files[1];
            std::cout << "Device does not support images. Using buffers instead of image2D structures." << std::endl;
        } else {
            kernel_file = files[0];
            std::cout << "Device supports images." << std::endl;
        }
    } else {
        kernel_file = files[1];
        std::cout << "Bypass image support is True. Using buffers instead of image2D structures." << std::endl;
    }
    controller.SetImageSupport(image_support);

    *context = controller.CreateContext(platforms[m_opt.platform_index], devices);
    *command_queue = controller.CreateCommandQueue(*context, devices[m_opt.device_index]);
    *program = controller.CreateProgram(*context, devices[m_opt.device_index], kernel_file.c_str());
    *kernel = controller.CreateKernel(*program, kernel_name.c_str());
}

#ifdef MI355_NO_OPENCV
// binary PPM (P6, maxval 255) -> tightly packed RGBA, A = 255
static bool read_ppm_rgba(const std::string& path, std::vector<unsigned char>* rgba, cl_int* width, cl_int* height)
{
    std::ifstream f(path, std::ios::binary);
    std::string magic;
    int w = 0, h = 0, maxv = 0;
    auto next_int = [&](int& v) {
        std::string tok;
        while (f >> tok) {
            if (tok[0] == '#') {
                std::string rest;
                std::getline(f, rest);
                continue;
            }
            v = std::atoi(tok.c_str());
            return true;
        }
        return false;
    };
    if (!(f >> magic) || magic != "P6" || !next_int(w) || !next_int(h) || !next_int(maxv) || maxv != 255 ||
        w <= 0 || h <= 0)
        return false;
    f.get();  // the single whitespace byte after maxval
    std::vector<unsigned char> rgb((size_t)w * h * 3);
    if (!f.read(reinterpret_cast<char*>(rgb.data()), (std::streamsize)rgb.size()))
        return false;
    rgba->resize((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        (*rgba)[4 * i] = rgb[3 * i];
        (*rgba)[4 * i + 1] = rgb[3 * i + 1];
        (*rgba)[4 * i + 2] = rgb[3 * i + 2];
        (*rgba)[4 * i + 3] = 255;
    }
    *width = w;
    *height = h;
    return true;
}
#endif

void ProgramHandler::LoadFrame(std::string image_path, std::vector<unsigned char>* input_data, cl_int* width,
                                    cl_int* height, Logger& logger)
{
#ifdef MI355_NO_OPENCV
    if (!read_ppm_rgba(image_path, input_data, width, height))
        logger.log("Failed to load image", Logger::LogLevel::ERROR);
#else
    cv::Mat image = cv::imread(image_path, cv::IMREAD_COLOR);
    if (image.empty()) {
        logger.log("Failed to load image", Logger::LogLevel::ERROR);
        return;
    }
    if (m_opt.display_images)
        cv::imshow("Reference Image Window", image);
    cv::cvtColor(image, image, cv::COLOR_BGR2RGBA);
    *width = image.cols;
    *height = image.rows;
    input_data->assign(image.data, image.data + image.total() * 4);
#endif
}

static int method_index(const std::string& method)
{
    if (method == "EDGE")
        return 1;
    if (method == "GAUSSIAN")
        return 2;
    return 0;  // the reference treats everything else as GRAYSCALE (:171-175)
}

std::vector<unsigned char> ProgramHandler::PerformOpenCL(
    Controller& controller, std::string image_path, cl_context* context, cl_command_queue* command_queue,
    cl_kernel* kernel, double& avg_opencl_execution_time, double& avg_opencl_kernel_write_time,
    double& avg_opencl_kernel_execution_time, double& avg_opencl_kernel_read_time, double& avg_opencl_kernel_operation,
    cl_int& width, cl_int& height, Logger& logger, std::string method)
{
    std::vector<unsigned char> input_data;
    std::vector<unsigned char> function_output;
    width = height = 0;
    LoadFrame(image_path, &input_data, &width, &height, logger);
    const int which = method_index(method);
    const size_t npx = (size_t)width * height;

    double total_execution_time = 0.0, total_write_time = 0.0, total_kernel_time = 0.0, total_read_time = 0.0;
    for (int i = 0; i < m_opt.iterations; i++) {
        std::vector<cl_ulong> ev;
        const auto t0 = std::chrono::high_resolution_clock::now();
        switch (which) {
        case 0:
            function_output = std::vector<unsigned char>(npx * 4);
            controller.PerformCLImageGrayscaling(context, command_queue, kernel, &ev, &input_data, &function_output,
                                                 width, height, logger);
            break;
        case 1:
            function_output = std::vector<unsigned char>(npx);
            controller.PerformCLImageEdgeDetection(context, command_queue, kernel, &ev, &input_data,
                                                   &function_output, width, height, logger);
            break;
        default:
            function_output = std::vector<unsigned char>(npx * 4);
            controller.PerformCLGaussianBlur(m_opt.gauss_k, m_opt.gauss_sigma, context, command_queue, kernel,
                                             &ev, &input_data, &function_output, width, height, logger);
            break;
        }
        logger.log("Performing OpenCL " + method + " on " + image_path + "...", Logger::LogLevel::INFO);
        const auto t1 = std::chrono::high_resolution_clock::now();
        total_execution_time += std::chrono::duration<double, std::milli>(t1 - t0).count();
        if (ev.size() >= 6) {
            total_write_time += (ev[1] - ev[0]) * 1e-6;
            total_kernel_time += (ev[3] - ev[2]) * 1e-6;
            total_read_time += (ev[5] - ev[4]) * 1e-6;
            if (m_opt.log_events) {
                static const char* names[6] = {"Write event start: ", "Write event end: ", "Kernel event start: ",
                                               "Kernel event end: ", "Read event start: ",  "Read event end: "};
                for (int e = 0; e < 6; e++)
                    logger.log(names[e] + std::to_string(ev[e]), Logger::LogLevel::INFO);
            }
        }
    }
    logger.log("OpenCL " + method + " conversion complete", Logger::LogLevel::INFO);

    const double n = m_opt.iterations > 0 ? m_opt.iterations : 1;
    avg_opencl_execution_time = total_execution_time / n;
    avg_opencl_kernel_write_time = total_write_time / n;
    avg_opencl_kernel_execution_time = total_kernel_time / n;
    avg_opencl_kernel_read_time = total_read_time / n;
    avg_opencl_kernel_operation = (total_write_time + total_kernel_time + total_read_time) / n;
    return function_output;
}

std::vector<unsigned char> ProgramHandler::PerformOpenCL(Controller& controller, const cv::Mat& input_frame,
                                                         cl_context* context, cl_command_queue* command_queue,
                                                         cl_kernel* kernel, cl_int& width, cl_int& height,
                                                         Logger& logger, std::string method)
{
    assert(input_frame.cols == width && input_frame.rows == height);
    std::vector<cl_ulong> profiling_events;
    std::vector<unsigned char> function_output;
    // the frame must already be RGBA (RT/RealtimeImageProcessing.cpp:330)
    std::vector<unsigned char> input_data(input_frame.data, input_frame.data + input_frame.total() * 4);
    const size_t npx = (size_t)width * height;

    const auto t0 = std::chrono::high_resolution_clock::now();
    switch (method_index(method)) {
    case 0:
        function_output = std::vector<unsigned char>(npx * 4);
        logger.log("Performing OpenCL Grayscaling...", Logger::LogLevel::INFO);
        controller.PerformCLImageGrayscaling(context, command_queue, kernel, &profiling_events, &input_data,
                                             &function_output, width, height, logger);
        logger.log("OpenCL Grayscale conversion complete", Logger::LogLevel::INFO);
        break;
    case 1:
        function_output = std::vector<unsigned char>(npx);
        logger.log("Performing OpenCL Edge Detection...", Logger::LogLevel::INFO);
        controller.PerformCLImageEdgeDetection(context, command_queue, kernel, &profiling_events, &input_data,
                                               &function_output, width, height, logger);
        logger.log("OpenCL Edge Detection complete", Logger::LogLevel::INFO);
        break;
    default:
        function_output = std::vector<unsigned char>(npx * 4);
        logger.log("Performing OpenCL Gaussian Blur...", Logger::LogLevel::INFO);
        controller.PerformCLGaussianBlur(m_opt.gauss_k, m_opt.gauss_sigma, context, command_queue, kernel,
                                         &profiling_events, &input_data, &function_output, width, height, logger);
        logger.log("OpenCL Gaussian Blur complete", Logger::LogLevel::INFO);
        break;
    }
    const auto t1 = std::chrono::high_resolution_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    logger.log("OpenCL " + method + " execution time: " + std::to_string(ms) + " ms", Logger::LogLevel::INFO);
    return function_output;
}

// ---- MI355X extension: one batch over several GPUs (include/mi355_imgfilter.h "device group") ---------------------
ProgramHandler::~ProgramHandler()
{
    if (m_group)
        mi355_group_destroy(m_group);
}

std::vector<unsigned char> ProgramHandler::PerformOpenCLBatch(const std::vector<unsigned char>& rgba_frames, int nframes,
                                                              cl_int width, cl_int height, Logger& logger,
                                                              std::string method, std::vector<int> devices)
{
    auto fail = [&](const std::string& what, int rc) {
        logger.log("ERROR: " + what + " (" + std::string(mi355_strerror(rc)) + ")", Logger::LogLevel::ERROR);
        std::cerr << "ERROR: " << what << " (" << mi355_strerror(rc) << ")" << std::endl;
        exit(EXIT_FAILURE);
    };
    if (devices.empty()) {
        int n = 0;
        mi355_device_count(&n);
        for (int i = 0; i < n; i++)
            devices.push_back(i);
    }
    if (nframes <= 0 || width <= 0 || height <= 0 ||
        rgba_frames.size() < (size_t)nframes * (size_t)width * (size_t)height * 4)
        fail("PerformOpenCLBatch: frame buffer smaller than nframes x width x height x 4", MI355_ERR_BAD_ARG);
    if (!m_group || devices != m_group_devices) {
        if (m_group)
            mi355_group_destroy(m_group);
        m_group = nullptr;
        const int rc = mi355_group_create((int)devices.size(), devices.data(), &m_group);
        if (rc != MI355_OK)
            fail("PerformOpenCLBatch: no usable GPU group", rc);
        m_group_devices = devices;
    }
    const int filter = method == "PIPELINE" ? MI355_FILTER_PIPELINE
                                            : (method_index(method) == 1 ? MI355_FILTER_SOBEL
                                               : method_index(method) == 2 ? MI355_FILTER_GAUSS : MI355_FILTER_GRAY);
    std::vector<unsigned char> out((size_t)nframes * width * height * (size_t)mi355_filter_out_bpp(filter));
    double ms = 0.0;
    logger.log("Performing " + method + " on " + std::to_string(nframes) + " frames over " +
                   std::to_string(devices.size()) + " GPU(s)...", Logger::LogLevel::INFO);
    const int rc = mi355_group_filter_batched(m_group, filter, rgba_frames.data(), out.data(), width, height, nframes,
                                              m_opt.gauss_k, m_opt.gauss_sigma, &ms);
    if (rc != MI355_OK)
        fail("PerformOpenCLBatch: " + method + " failed", rc);
    logger.log(method + " batch execution time: " + std::to_string(ms) + " ms", Logger::LogLevel::INFO);
    return out;
}
