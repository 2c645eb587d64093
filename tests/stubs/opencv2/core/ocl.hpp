// TEST-ONLY declaration stub (see ../opencv.hpp): cv::ocl::setUseOpenCL, RT/RealtimeImageProcessing.cpp:431.
#ifndef MI355_TEST_OPENCV_OCL_STUB_HPP
#define MI355_TEST_OPENCV_OCL_STUB_HPP
#include <opencv2/opencv.hpp>
namespace cv { namespace ocl { void setUseOpenCL(bool flag); } }
#endif
