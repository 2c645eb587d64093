// Comparator.hpp — present because the app includes it (RT/RealtimeImageProcessing.cpp:7); the class is
// never instantiated there.  Declaration only: the CPU comparison path is NOT part of this
// library — there is no CPU implementation of any filter in the product.  The CPU restatement used to
// check parity lives in oracle/ as test infrastructure.
#ifndef COMPARATOR_H
#define COMPARATOR_H

#include <Logger.hpp>

#include <string>

#if __has_include(<opencv2/opencv.hpp>)
#include <opencv2/opencv.hpp>
#else
#include <cv_min.hpp>
#endif

class Comparator
{
public:
    Comparator(int num_methods, int num_iterations);

    cv::Mat PerformCPU_Grayscaling(std::string image_path, double& avg_cpu_execution_time, Logger& logger);

private:
    int m_num_methods;
    int NUMBER_OF_ITERATIONS;

    // include/Comparator.hpp:21 of the reference (mean absolute difference of two images); declaration only,
    // like the rest of this class
    double ComputeMAE(const cv::Mat& reference, const cv::Mat& result, Logger& logger);
};

#endif  // COMPARATOR_H
