// FileHandler.hpp — the app includes it (RT/RealtimeImageProcessing.cpp:6) and default-constructs one, so
// the class keeps the reference's public surface (include/FileHandler.hpp:13-18).  I/O convenience only,
// outside the hot path; the CSV header is the reference's (RT/src/FileHandler.cpp:28).
#ifndef FILEHANDLER_H
#define FILEHANDLER_H

#include <filesystem>
#include <fstream>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#if __has_include(<opencv2/opencv.hpp>)
#include <opencv2/opencv.hpp>
#else
#include <cv_min.hpp>
#endif

namespace fs = std::filesystem;

class FileHandler
{
public:
    FileHandler();

    std::vector<std::string> LoadImages(const std::string& directory);

    void SaveImages(std::string image_path, cv::Mat& opencl_output_image);
    void WriteResultsToCSV(const std::string& filename,
                           std::vector<std::tuple<std::string, std::string, std::string, int, double, double, double,
                                                  double, double, double, double>>& results);

private:
    bool SAVE_IMAGES;

    std::string m_directory_name;
    std::vector<std::string> m_image_paths;
};

#endif  // FILEHANDLER_H
