// InfoPlatform.hpp — platform report helper.  The reference's Controller.hpp includes a header of this name
// (include/InfoPlatform.hpp:9-23 there) and Controller::DisplayPlatformInformation builds one, so the type and
// its four public members exist here too; what they print is what this library's clGetPlatformInfo answers.
// OpenCL introspection as such is outside the rebuilt path.
#pragma once

#include <string>
#include <iostream>
#include <CL/cl.h>

class InfoPlatform
{
    // filled once at construction
    struct Strings {
        std::string profile, name, version, vendor;
    } m_info;

public:
    explicit InfoPlatform(cl_platform_id id);

    // one of CL_PLATFORM_PROFILE / _NAME / _VERSION / _VENDOR; anything else gives an empty string
    std::string GetPlatformInfo(cl_platform_info name);
    // all four, one per line, on stdout
    void Display();
    // "\t<str>:\t<value>" for one query, asked from the library again
    void DisplaySinglePlatformInfo(cl_platform_id id, cl_platform_info name, std::string str);
};
