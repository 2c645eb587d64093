/*
 * CL/cl.h — NOT the Khronos header.  A minimal, self-written stand-in exposing only the OpenCL *names*
 * that the reference's callers of Controller / ProgramHandler touch directly, so that
 * RealtimeImageProcessing.cpp and ProgramHandler-style code compile unchanged against this library with
 * no libOpenCL and no Khronos headers installed (SURVEY.md §8b "Types").
 *
 * Handles are opaque pointers to small records owned by libmi355_host; nothing here talks to an OpenCL
 * runtime.  The reference vendors the real headers under include/CL (out of scope: SURVEY.md §2 row 12).
 *
 * Names used by the unchanged callers:
 *   types      cl_int cl_uint cl_ulong cl_bool cl_platform_id cl_device_id cl_context cl_command_queue
 *              cl_program cl_kernel cl_mem cl_sampler cl_event cl_platform_info cl_device_info
 *   constants  CL_SUCCESS CL_TRUE CL_FALSE CL_DEVICE_NAME CL_DEVICE_IMAGE_SUPPORT CL_PLATFORM_*
 *   functions  clGetDeviceInfo clGetPlatformInfo clReleaseKernel clReleaseProgram clReleaseCommandQueue
 *              clReleaseContext clReleaseMemObject clReleaseSampler
 *              (RT/RealtimeImageProcessing.cpp:282-285,423-426; RT/src/ProgramHandler.cpp:62,85;
 *               RT/src/InfoPlatform.cpp:65,75; RT/src/Controller.cpp:199-232)
 *
 * No image entry points (clCreateImage2D, clEnqueueReadImage, clEnqueueWriteImage) exist here:
 * clGetDeviceInfo(CL_DEVICE_IMAGE_SUPPORT) answers CL_FALSE unless the host process opts in with
 * MI355_CL_IMAGE_SUPPORT=1, and even then only the bundled Controller's image2d_t code path is backed
 * (it calls mi355_image2d_rgba8 directly) — code that trusts the capability bit to create cl images
 * itself would not link (INTEGRATION.md "image2d_t mode").
 */
#ifndef MI355_CL_SHAPED_H
#define MI355_CL_SHAPED_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t cl_int;
typedef uint32_t cl_uint;
typedef uint64_t cl_ulong;
typedef cl_uint cl_bool;
typedef cl_uint cl_platform_info;
typedef cl_uint cl_device_info;
typedef cl_ulong cl_bitfield;
typedef cl_bitfield cl_command_queue_properties;

typedef struct _cl_platform_id* cl_platform_id;
typedef struct _cl_device_id* cl_device_id;
typedef struct _cl_context* cl_context;
typedef struct _cl_command_queue* cl_command_queue;
typedef struct _cl_program* cl_program;
typedef struct _cl_kernel* cl_kernel;
typedef struct _cl_mem* cl_mem;
typedef struct _cl_sampler* cl_sampler;
typedef struct _cl_event* cl_event;

#define CL_SUCCESS 0
#define CL_DEVICE_NOT_FOUND (-1)
#define CL_INVALID_VALUE (-30)
#define CL_INVALID_DEVICE (-33)
#define CL_INVALID_KERNEL_NAME (-46)
#define CL_FALSE 0
#define CL_TRUE 1

#define CL_PLATFORM_PROFILE 0x0900
#define CL_PLATFORM_VERSION 0x0901
#define CL_PLATFORM_NAME 0x0902
#define CL_PLATFORM_VENDOR 0x0903
#define CL_DEVICE_IMAGE_SUPPORT 0x1016
#define CL_DEVICE_NAME 0x102B
#define CL_QUEUE_PROFILING_ENABLE (1 << 1)

cl_int clGetPlatformInfo(cl_platform_id platform, cl_platform_info name, size_t size, void* value,
                         size_t* size_ret);
cl_int clGetDeviceInfo(cl_device_id device, cl_device_info name, size_t size, void* value, size_t* size_ret);
cl_int clReleaseKernel(cl_kernel kernel);
cl_int clReleaseProgram(cl_program program);
cl_int clReleaseCommandQueue(cl_command_queue queue);
cl_int clReleaseContext(cl_context context);
cl_int clReleaseMemObject(cl_mem mem);
cl_int clReleaseSampler(cl_sampler sampler);

#ifdef __cplusplus
}
#endif
#endif /* MI355_CL_SHAPED_H */
