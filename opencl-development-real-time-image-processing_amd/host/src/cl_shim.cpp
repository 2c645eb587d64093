// cl_shim.cpp — the handful of cl* C symbols the reference's callers reference directly
// (RT/RealtimeImageProcessing.cpp:282-285,423-426; RT/src/ProgramHandler.cpp:62,85;
// RT/src/InfoPlatform.cpp:65,75), implemented over the records in handles.hpp.  No OpenCL runtime.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "handles.hpp"

namespace mi355_host {

static _cl_platform_id g_platform = {0};
static _cl_device_id g_devices[64];
static int g_ndev = -1;

cl_platform_id the_platform() { return &g_platform; }

int device_count()
{
    if (g_ndev < 0) {
        int n = 0;
        if (mi355_device_count(&n) != MI355_OK)
            n = 0;
        if (n > 64)
            n = 64;
        for (int i = 0; i < n; i++)
            g_devices[i].index = i;
        g_ndev = n;
    }
    return g_ndev;
}

cl_device_id device_handle(int index)
{
    return (index >= 0 && index < device_count()) ? &g_devices[index] : nullptr;
}

static cl_int put_string(const char* s, size_t size, void* value, size_t* size_ret)
{
    const size_t need = std::strlen(s) + 1;
    if (size_ret)
        *size_ret = need;
    if (value) {
        if (size < need)
            return CL_INVALID_VALUE;
        std::memcpy(value, s, need);
    }
    return CL_SUCCESS;
}

}  // namespace mi355_host

extern "C" {

#define HOST_API __attribute__((visibility("default")))

HOST_API cl_int clGetPlatformInfo(cl_platform_id platform, cl_platform_info name, size_t size, void* value,
                                  size_t* size_ret)
{
    if (platform != mi355_host::the_platform())
        return CL_INVALID_VALUE;
    switch (name) {
    case CL_PLATFORM_PROFILE: return mi355_host::put_string("FULL_PROFILE", size, value, size_ret);
    case CL_PLATFORM_NAME: return mi355_host::put_string("MI355X HIP image-filter path", size, value, size_ret);
    case CL_PLATFORM_VERSION: return mi355_host::put_string(mi355_build_info(), size, value, size_ret);
    case CL_PLATFORM_VENDOR: return mi355_host::put_string("Advanced Micro Devices, Inc.", size, value, size_ret);
    default: return CL_INVALID_VALUE;
    }
}

HOST_API cl_int clGetDeviceInfo(cl_device_id device, cl_device_info name, size_t size, void* value,
                                size_t* size_ret)
{
    if (!device || device != mi355_host::device_handle(device->index))
        return CL_INVALID_DEVICE;
    if (name == CL_DEVICE_NAME) {
        char buf[256];
        std::snprintf(buf, sizeof(buf), "AMD Instinct MI355X (gfx950) #%d", device->index);
        return mi355_host::put_string(buf, size, value, size_ret);
    }
    if (name == CL_DEVICE_IMAGE_SUPPORT) {
        // CL_FALSE: an application that does not bypass image support (no shipped one does,
        // RT/RealtimeImageProcessing.cpp:23) then takes its own "device does not support images" branch
        // (RT/src/ProgramHandler.cpp:78-84) and stays on the buffer kernels — the 0.64-0.75-of-roofline path with the
        // CPU path's pixel values.  The image2d_t semantics exist (csrc/image2d.hip, mi355_image2d_rgba8) but they
        // compute DIFFERENT pixels (zero borders, image-mode table, red-channel Sobel) at a lower rate
        // (INTEGRATION.md "image2d_t mode"), so they are an explicit opt-in: MI355_CL_IMAGE_SUPPORT=1 in the
        // environment of the host process, or Controller::SetImageSupport(CL_TRUE) after InitOpenCL.  This shim
        // exports no clCreateImage2D / clEnqueue{Read,Write}Image: only the Controller-level image path is backed.
        if (size_ret)
            *size_ret = sizeof(cl_bool);
        if (value) {
            if (size < sizeof(cl_bool))
                return CL_INVALID_VALUE;
            const char* e = std::getenv("MI355_CL_IMAGE_SUPPORT");
            *static_cast<cl_bool*>(value) = (e && e[0] == '1') ? CL_TRUE : CL_FALSE;
        }
        return CL_SUCCESS;
    }
    return CL_INVALID_VALUE;
}

HOST_API cl_int clReleaseKernel(cl_kernel kernel)
{
    delete kernel;
    return CL_SUCCESS;
}

HOST_API cl_int clReleaseProgram(cl_program program)
{
    delete program;
    return CL_SUCCESS;
}

HOST_API cl_int clReleaseCommandQueue(cl_command_queue queue)
{
    if (queue) {
        if (queue->ctx)
            mi355_ctx_destroy(queue->ctx);
        delete queue;
    }
    return CL_SUCCESS;
}

HOST_API cl_int clReleaseContext(cl_context context)
{
    delete context;
    return CL_SUCCESS;
}

HOST_API cl_int clReleaseMemObject(cl_mem mem)
{
    delete mem;
    return CL_SUCCESS;
}

HOST_API cl_int clReleaseSampler(cl_sampler sampler)
{
    delete sampler;
    return CL_SUCCESS;
}

}  // extern "C"
