// handles.hpp — the records behind the opaque cl_* handle types of include/CL/cl.h.
#pragma once
#include <CL/cl.h>

#include "../../../include/mi355_imgfilter.h"

struct _cl_platform_id {
    int index;
};
struct _cl_device_id {
    int index;  // HIP device ordinal
};
struct _cl_context {
    int num_devices;
};
struct _cl_command_queue {
    mi355_ctx* ctx;  // one GPU + one HIP stream + pooled buffers
    int device;
};
enum { MI355_FAMILY_GRAY = 0, MI355_FAMILY_EDGE = 1, MI355_FAMILY_GAUSS = 2 };
struct _cl_program {
    int family;
};
struct _cl_kernel {
    int family;
};
struct _cl_mem {
    int unused;
};
struct _cl_sampler {
    int unused;
};

namespace mi355_host {
cl_platform_id the_platform();
cl_device_id device_handle(int index);  // nullptr when out of range
int device_count();
}  // namespace mi355_host
