// host_app.cpp — drives the C++ boundary exactly the way the reference's RealtimeImageProcessing.cpp does
// (RT/RealtimeImageProcessing.cpp:435-442 setup, :351-355 / :372-376 / :398-402 per-filter InitOpenCL +
// PerformOpenCL(frame), :423-426 raw clRelease*), on raw RGBA frames instead of a camera, and dumps the
// outputs so tests/test_host_cpp.py can compare them with the oracle.
//
//   host_app <in.rgba> <width> <height> <out_prefix> [<image.ppm>]
// writes <out_prefix>.gray (w*h*4), .edge (w*h), .gauss (w*h*4), .gauss17 (w*h*4, the ProgramHandler
// default k=17 sigma=6), .weights (25 floats), .prof (6 u64 from one Controller call) and, when a PPM is
// given, .ppm_gray from the N-iteration PerformOpenCL(image_path, ...) overload; .batch_pipe (w*h) = first frame of a
// five-frame PerformOpenCLBatch("PIPELINE") over a two-member group.
#include <ProgramHandler.hpp>
#include <FileHandler.hpp>
#include <Comparator.hpp>

#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <fstream>

static std::vector<std::string> GRAYSCALE_KERNELS = {"grayscale_images.cl", "grayscale_base.cl"};
static std::vector<std::string> GAUSSIAN_KERNELS = {"gaussian_images.cl", "gaussian_base.cl"};
static std::vector<std::string> EDGE_KERNELS = {"edge_images.cl", "edge_base.cl"};

static void dump(const std::string& path, const void* p, size_t n)
{
    std::ofstream f(path, std::ios::binary);
    f.write(static_cast<const char*>(p), (std::streamsize)n);
}

static std::vector<unsigned char> run(ProgramHandler& ph, Controller& controller, Logger& logger,
                                      const cv::Mat& frame, cl_int w, cl_int h, const std::string& method)
{
    cl_context context;
    cl_command_queue command_queue;
    cl_program program;
    cl_kernel kernel;
    ph.InitOpenCL(controller, &context, &command_queue, &program, &kernel, method, logger);
    auto out = ph.PerformOpenCL(controller, frame, &context, &command_queue, &kernel, w, h, logger, method);
    clReleaseKernel(kernel);
    clReleaseProgram(program);
    clReleaseCommandQueue(command_queue);
    clReleaseContext(context);
    return out;
}

int main(int argc, char** argv)
{
    if (argc < 5) {
        std::fprintf(stderr, "usage: host_app in.rgba w h out_prefix [image.ppm]\n");
        return 2;
    }
    const cl_int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
    const std::string prefix = argv[4];
    std::vector<unsigned char> rgba((size_t)w * h * 4);
    {
        std::ifstream f(argv[1], std::ios::binary);
        if (!f.read(reinterpret_cast<char*>(rgba.data()), (std::streamsize)rgba.size())) {
            std::fprintf(stderr, "short read\n");
            return 2;
        }
    }
#ifdef MI355_NO_OPENCV
    cv::Mat frame(h, w, 4, rgba.data());
#else
    cv::Mat frame(h, w, CV_8UC4, rgba.data());
#endif

    Logger& logger = Logger::getInstance();
    Controller controller;
    FileHandler file_handler;
    (void)file_handler;

    // the app's constructor call (5 flags, default k=17 sigma=6) and a 5x5 one
    ProgramHandler ph17(1, false, false, false, true);
    ProgramHandler ph5(1, false, false, false, true, 5, 1.5f);
    for (ProgramHandler* ph : {&ph17, &ph5}) {
        ph->InitLogger(logger, Logger::LogLevel::ERROR, false);
        ph->SetDeviceProperties(0, 0);
        ph->AddKernels(GRAYSCALE_KERNELS, "GRAYSCALE");
        ph->AddKernels(EDGE_KERNELS, "EDGE");
        ph->AddKernels(GAUSSIAN_KERNELS, "GAUSSIAN");
    }

    auto gray = run(ph5, controller, logger, frame, w, h, "GRAYSCALE");
    auto edge = run(ph5, controller, logger, frame, w, h, "EDGE");
    auto gauss = run(ph5, controller, logger, frame, w, h, "GAUSSIAN");
    auto gauss17 = run(ph17, controller, logger, frame, w, h, "GAUSSIAN");
    if (controller.GetImageSupport() != CL_FALSE)
        return 3;
    dump(prefix + ".gray", gray.data(), gray.size());
    dump(prefix + ".edge", edge.data(), edge.size());
    dump(prefix + ".gauss", gauss.data(), gauss.size());
    dump(prefix + ".gauss17", gauss17.data(), gauss17.size());

    auto weights = controller._GenerateGausianKernel(5, 1.5f);
    dump(prefix + ".weights", weights.data(), weights.size() * sizeof(float));

    // one direct Controller call: profiling contract (six appended timestamps per call)
    {
        cl_context context;
        cl_command_queue queue;
        cl_program program;
        cl_kernel kernel;
        ph5.InitOpenCL(controller, &context, &queue, &program, &kernel, "EDGE", logger);
        std::vector<cl_ulong> ev;
        std::vector<unsigned char> out((size_t)w * h);
        cl_int ww = w, hh = h;
        controller.PerformCLImageEdgeDetection(&context, &queue, &kernel, &ev, &rgba, &out, ww, hh, logger);
        controller.PerformCLImageEdgeDetection(&context, &queue, &kernel, &ev, &rgba, &out, ww, hh, logger);
        if (ev.size() != 12 || out != edge)
            return 4;
        dump(prefix + ".prof", ev.data(), ev.size() * sizeof(cl_ulong));
        controller.Cleanup(context, queue, program, kernel);
    }

    // MI355X extension: the same frame five times as one batch over two group members bound to GPU 0 (own thread,
    // own stream each) — every frame's result must be the per-frame call's
    {
        std::vector<unsigned char> batch;
        for (int f = 0; f < 5; f++)
            batch.insert(batch.end(), rgba.begin(), rgba.end());
        auto be = ph5.PerformOpenCLBatch(batch, 5, w, h, logger, "EDGE", {0, 0});
        auto bg = ph5.PerformOpenCLBatch(batch, 5, w, h, logger, "GAUSSIAN", {0, 0});
        auto bp = ph5.PerformOpenCLBatch(batch, 5, w, h, logger, "PIPELINE", {0, 0});
        if (be.size() != edge.size() * 5 || bg.size() != gauss.size() * 5 || bp.size() != edge.size() * 5)
            return 6;
        for (int f = 0; f < 5; f++) {
            if (!std::equal(edge.begin(), edge.end(), be.begin() + (long)f * (long)edge.size()) ||
                !std::equal(gauss.begin(), gauss.end(), bg.begin() + (long)f * (long)gauss.size()))
                return 7;
        }
        dump(prefix + ".batch_pipe", bp.data(), edge.size());
    }

    if (argc > 5) {
        cl_context context;
        cl_command_queue queue;
        cl_program program;
        cl_kernel kernel;
        ProgramHandler ph(3, false, false, false, true, 5, 1.5f);
        ph.SetDeviceProperties(0, 0);
        ph.InitOpenCL(controller, &context, &queue, &program, &kernel, "GRAYSCALE", logger);
        double t_exec = 0, t_write = 0, t_kernel = 0, t_read = 0, t_op = 0;
        cl_int iw = 0, ih = 0;
        auto out = ph.PerformOpenCL(controller, std::string(argv[5]), &context, &queue, &kernel, t_exec, t_write,
                                    t_kernel, t_read, t_op, iw, ih, logger, "GRAYSCALE");
        if (t_kernel <= 0 || t_op < t_kernel || iw <= 0)
            return 5;
        dump(prefix + ".ppm_gray", out.data(), out.size());
        std::printf("image %dx%d e2e %.3f ms write %.4f kernel %.4f read %.4f\n", iw, ih, t_exec, t_write, t_kernel,
                    t_read);
        controller.Cleanup(context, queue, program, kernel);
    }
    // BYPASS_IMAGE_SUPPORT = false (no shipped app): InitOpenCL probes CL_DEVICE_IMAGE_SUPPORT, picks
    // the *_images.cl kernel file and switches the Controller to image2d_t semantics
    {
        ProgramHandler phi(1, false, false, false, false, 5, 1.5f);
        phi.InitLogger(logger, Logger::LogLevel::ERROR, false);
        phi.SetDeviceProperties(0, 0);
        phi.AddKernels(GRAYSCALE_KERNELS, "GRAYSCALE");
        phi.AddKernels(EDGE_KERNELS, "EDGE");
        phi.AddKernels(GAUSSIAN_KERNELS, "GAUSSIAN");
        Controller image_controller;
        auto igray = run(phi, image_controller, logger, frame, w, h, "GRAYSCALE");
        // the device reports CL_DEVICE_IMAGE_SUPPORT = CL_FALSE unless the host process opted in
        // (MI355_CL_IMAGE_SUPPORT=1): without the opt-in a BYPASS = false application stays on the buffer path
        const char* optin = std::getenv("MI355_CL_IMAGE_SUPPORT");
        if (!(optin && optin[0] == '1')) {
            if (image_controller.GetImageSupport() != CL_FALSE || igray != gray)
                return 8;
            std::printf("host_app ok (image support not opted in)\n");
            return 0;
        }
        if (image_controller.GetImageSupport() != CL_TRUE)
            return 6;
        auto iedge = run(phi, image_controller, logger, frame, w, h, "EDGE");
        auto igauss = run(phi, image_controller, logger, frame, w, h, "GAUSSIAN");
        dump(prefix + ".img_gray", igray.data(), igray.size());
        dump(prefix + ".img_edge", iedge.data(), iedge.size());
        dump(prefix + ".img_gauss", igauss.data(), igauss.size());
        auto iw = image_controller._GenerateGausianKernel(5, 1.5f);
        dump(prefix + ".img_weights", iw.data(), iw.size() * sizeof(float));
    }
    std::printf("host_app ok\n");
    return 0;
}
