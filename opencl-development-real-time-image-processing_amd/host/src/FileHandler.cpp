// FileHandler.cpp — directory scan and the results CSV (reference: RT/src/FileHandler.cpp:1-34).
#include "FileHandler.hpp"

#include <algorithm>

FileHandler::FileHandler() : SAVE_IMAGES{false}, m_directory_name{"images"}, m_image_paths{} {}

std::vector<std::string> FileHandler::LoadImages(const std::string& directory)
{
    std::error_code ec;
    for (const auto& entry : fs::directory_iterator(directory, ec)) {
        const auto ext = entry.path().extension();
        if (ext == ".jpg" || ext == ".png" || ext == ".ppm")
            m_image_paths.push_back(entry.path().string());
    }
    std::sort(m_image_paths.begin(), m_image_paths.end());
    return m_image_paths;
}

void FileHandler::SaveImages(std::string image_path, cv::Mat& opencl_output_image)
{
#ifndef MI355_NO_OPENCV
    if (SAVE_IMAGES)
        cv::imwrite("images/opencl_grayscale_" + fs::path(image_path).filename().string(), opencl_output_image);
#else
    (void)image_path;
    (void)opencl_output_image;
#endif
}

void FileHandler::WriteResultsToCSV(
    const std::string& filename,
    std::vector<std::tuple<std::string, std::string, std::string, int, double, double, double, double, double, double,
                           double>>& results)
{
    std::ofstream file(filename);
    file << "Timestamp, Image, Resolution, Num_Iterations, avg_CPU_Time_ms, avg_OpenCL_Time_ms, "
            "avg_OpenCL_kernel_ms, avg_OpenCL_kernel_write_ms, avg_OpenCL_kernel_read_ms, "
            "avg_OpenCL_kernel_operation_ms, Error_MAE\n";
    for (const auto& [timestamp, image, resolution, iters, cpu, ocl, kern, wr, rd, op, mae] : results)
        file << timestamp << ", " << image << ", " << resolution << ", " << iters << ", " << cpu << ", " << ocl
             << ", " << kern << ", " << wr << ", " << rd << ", " << op << ", " << mae << "\n";
}
