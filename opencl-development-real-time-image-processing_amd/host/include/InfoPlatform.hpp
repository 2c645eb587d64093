// InfoPlatform.hpp — same public surface as the reference's include/InfoPlatform.hpp:9-23 (it is named
// in Controller.hpp's include list, so it must exist); prints the four platform strings that
// clGetPlatformInfo of this library reports.  OpenCL introspection itself is out of scope.
#ifndef INFOPLATFORM_H
#define INFOPLATFORM_H

#include <CL/cl.h>
#include <iostream>
#include <string>

class InfoPlatform
{
public:
    InfoPlatform(cl_platform_id id);
    void DisplaySinglePlatformInfo(cl_platform_id id, cl_platform_info name, std::string str);
    void Display();

    std::string GetPlatformInfo(cl_platform_info name);

private:
    std::string m_profile, m_name, m_version, m_vendor;
};

#endif  // INFOPLATFORM_H
