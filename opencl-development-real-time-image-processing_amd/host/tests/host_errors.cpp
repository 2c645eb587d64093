// host_errors.cpp — the error conventions of the C++ boundary (SURVEY.md §8b "Error conventions"), one
// scenario per invocation so tests/test_host_cpp.py can check exit codes and messages:
//   unknown_method   ProgramHandler::InitOpenCL(..."BLUR"...)  -> "Unrecognised method", exit(1)
//                    (RT/src/ProgramHandler.cpp:75-78)
//   bad_kernel       Controller::CreateKernel(program, "nope") -> "Error: clCreateKernel (-46)", EXIT_FAILURE
//                    (RT/src/Controller.cpp:5-11,186-187)
//   bad_program      Controller::CreateProgram(ctx, dev, "foo.cl") -> NULL, message on stderr   (:138-148)
//   short_input      PerformCLImageGrayscaling with an undersized input vector -> ERROR logged, output
//                    untouched, no profiling events appended, process continues                 (:461-463)
//   even_kernel      PerformCLGaussianBlur with kernel_size 4 -> ERROR logged, call returns
//   log_throw        Logger::setLogFile on an unwritable path -> std::runtime_error             (RT/src/Logger.cpp:50-52)
//   csv <file>       FileHandler::WriteResultsToCSV: the reference's header and row format      (RT/src/FileHandler.cpp:25-34)
//   load_images <d>  FileHandler::LoadImages: the directory scan                                 (RT/src/FileHandler.cpp:5-14)
// Scenarios that need a device (all but log_throw, csv, load_images) are run on the GPU box only.
#include <FileHandler.hpp>
#include <ProgramHandler.hpp>

#include <cstdio>
#include <cstring>
#include <stdexcept>

int main(int argc, char** argv)
{
    if (argc < 2)
        return 2;
    const std::string what = argv[1];
    Logger& logger = Logger::getInstance();
    if (what == "log_throw") {
        try {
            logger.setLogFile("/nonexistent-dir/x/y.log", true);
        } catch (const std::runtime_error& e) {
            std::printf("caught: %s\n", e.what());
            return 0;
        }
        return 1;
    }
    if (what == "csv" && argc > 2) {
        FileHandler fh;
        std::vector<std::tuple<std::string, std::string, std::string, int, double, double, double, double, double, double,
                               double>>
            rows;
        rows.emplace_back("2026-10-04 12:00:00", "a.jpg", "640x512", 100, 0.30868, 1.512, 0.05, 0.7, 0.4, 1.15, 0.000410156);
        rows.emplace_back("2026-10-04 12:00:01", "b.jpg", "75x75", 3, 1.0, 2.0, 0.5, 0.25, 0.125, 0.875, 0.0);
        fh.WriteResultsToCSV(argv[2], rows);
        return 0;
    }
    if (what == "load_images" && argc > 2) {
        FileHandler fh;
        for (const auto& p : fh.LoadImages(argv[2]))
            std::printf("%s\n", p.c_str());
        return 0;
    }
    logger.setLogLevel(Logger::LogLevel::ERROR);
    logger.setTerminalDisplay(true);
    Controller controller;
    ProgramHandler ph(1, false, false, false, true, 5, 1.5f);
    ph.SetDeviceProperties(0, 0);
    cl_context context;
    cl_command_queue queue;
    cl_program program;
    cl_kernel kernel;
    if (what == "unknown_method") {
        ph.InitOpenCL(controller, &context, &queue, &program, &kernel, "BLUR", logger);
        return 0;  // not reached
    }
    auto platforms = controller.GetPlatforms();
    auto devices = controller.GetDevices(platforms[0]);
    context = controller.CreateContext(platforms[0], devices);
    queue = controller.CreateCommandQueue(context, devices[0]);
    if (what == "bad_program") {
        program = controller.CreateProgram(context, devices[0], "foo.cl");
        std::printf("program is %s\n", program == NULL ? "NULL" : "set");
        return program == NULL ? 0 : 1;
    }
    if (what == "bad_kernel") {
        program = controller.CreateProgram(context, devices[0], "gaussian_base.cl");
        kernel = controller.CreateKernel(program, "nope");
        return 0;  // not reached
    }
    cl_int w = 16, h = 8;
    if (what == "short_input") {
        program = controller.CreateProgram(context, devices[0], "grayscale_base.cl");
        kernel = controller.CreateKernel(program, "grayscale");
        std::vector<unsigned char> in(10), out((size_t)w * h * 4, 7);
        std::vector<cl_ulong> ev;
        controller.PerformCLImageGrayscaling(&context, &queue, &kernel, &ev, &in, &out, w, h, logger);
        const bool untouched = out.size() == (size_t)w * h * 4 && out[0] == 7 && ev.empty();
        std::printf("still running, output %s\n", untouched ? "untouched" : "modified");
        return untouched ? 0 : 1;
    }
    if (what == "even_kernel") {
        program = controller.CreateProgram(context, devices[0], "gaussian_base.cl");
        kernel = controller.CreateKernel(program, "gaussian_blur");
        std::vector<unsigned char> in((size_t)w * h * 4, 9), out((size_t)w * h * 4, 7);
        std::vector<cl_ulong> ev;
        int k = 4;
        float sigma = 1.0f;
        controller.PerformCLGaussianBlur(k, sigma, &context, &queue, &kernel, &ev, &in, &out, w, h, logger);
        std::printf("still running, %zu events\n", ev.size());
        return ev.empty() ? 0 : 1;
    }
    return 2;
}
