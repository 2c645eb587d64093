// InfoPlatform.cpp — prints the four platform strings (reference: RT/src/InfoPlatform.cpp:3-25).
#include "InfoPlatform.hpp"

static std::string query(cl_platform_id id, cl_platform_info name)
{
    size_t n = 0;
    if (clGetPlatformInfo(id, name, 0, nullptr, &n) != CL_SUCCESS || n == 0)
        return {};
    std::string s(n, '\0');
    if (clGetPlatformInfo(id, name, n, s.data(), nullptr) != CL_SUCCESS)
        return {};
    s.resize(n - 1);
    return s;
}

InfoPlatform::InfoPlatform(cl_platform_id id)
    : m_info{query(id, CL_PLATFORM_PROFILE), query(id, CL_PLATFORM_NAME), query(id, CL_PLATFORM_VERSION),
             query(id, CL_PLATFORM_VENDOR)}
{
}

void InfoPlatform::DisplaySinglePlatformInfo(cl_platform_id id, cl_platform_info name, std::string str)
{
    std::cout << "\t" << str << "\t" << query(id, name) << std::endl;
}

void InfoPlatform::Display()
{
    std::cout << "\nPLATFORM PROPERTIES:" << std::endl;
    std::cout << "\tCL_PLATFORM_PROFILE\t" << m_info.profile << std::endl;
    std::cout << "\tCL_PLATFORM_NAME\t" << m_info.name << std::endl;
    std::cout << "\tCL_PLATFORM_VERSION\t" << m_info.version << std::endl;
    std::cout << "\tCL_PLATFORM_VENDOR\t" << m_info.vendor << std::endl;
}

std::string InfoPlatform::GetPlatformInfo(cl_platform_info name)
{
    switch (name) {
    case CL_PLATFORM_PROFILE: return m_info.profile;
    case CL_PLATFORM_NAME: return m_info.name;
    case CL_PLATFORM_VERSION: return m_info.version;
    case CL_PLATFORM_VENDOR: return m_info.vendor;
    default: std::cerr << "Unrecognised platform info" << std::endl; return {};
    }
}
