#include "common.hpp"

namespace ogl {

std::string &last_error()
{
    static thread_local std::string msg;
    return msg;
}

int fail(int status, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return status;
}

}  // namespace ogl

extern "C" const char *ogl_last_error(void) { return ogl::last_error().c_str(); }
extern "C" int ogl_abi_version(void) { return OGL_AMD_ABI_VERSION; }
