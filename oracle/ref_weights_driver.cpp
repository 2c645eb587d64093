// ref_weights_driver.cpp — TEST INFRASTRUCTURE ONLY (see oracle/imgfilter_oracle.c header).
//
// A C-ABI door into the REFERENCE's own Gaussian weight generator, so the oracle's
// restatement can be pinned bit-for-bit against the real thing.  This file contains
// no reference code: it includes the reference's header where it lies and is linked
// against the reference's own Controller.cpp / Logger.cpp / InfoPlatform.cpp compiled
// from /root/reference by oracle/Makefile (target `ref`).  Those three sources need
// only the Khronos headers the reference vendors (include/CL) and the image's
// libOpenCL ICD loader — no OpenCV — so this part of the reference IS buildable here.
// The CPU filter loops are not (they live in mains that include <opencv2/opencv.hpp>).
//
// Reference entry used: Controller::_GenerateGausianKernel
//   (/root/reference/src/GaussianBlur/include/Controller.hpp:28, public there because
//    src/GaussianBlur/GaussianBlur.cpp:230 calls it from the CPU path),
// which dispatches on m_image_support to _GenerateGaussianKernelBuffers
//   (/root/reference/src/GaussianBlur/src/Controller.cpp:342-362) or ..Image2D (:364-393).
#include <Controller.hpp>

#include <cstring>

extern "C" __attribute__((visibility("default")))
int ref_gauss_weights(int k, float sigma, int image_support, float* out)
{
    Controller controller;
    // GaussianBlur.cpp sets this through InitOpenCL before the CPU path runs
    // (BYPASS_IMAGE_SUPPORT=true -> CL_FALSE); the constructor leaves it uninitialised.
    controller.SetImageSupport(image_support ? CL_TRUE : CL_FALSE);
    std::vector<float> w = controller._GenerateGausianKernel(k, sigma);
    std::memcpy(out, w.data(), w.size() * sizeof(float));
    return static_cast<int>(w.size());
}
