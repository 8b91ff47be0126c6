#include "common.hpp"

#include <cstdlib>

#include <dlfcn.h>

#include <mutex>

namespace ogl {

namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
};
const Roctx &roctx()
{
    static Roctx api;
    static std::once_flag once;
    std::call_once(once, [] {
        // rocprofv3 listens to the rocprofiler-sdk flavour of ROCTx; libroctx64 is the older one
        for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so",
                                 "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1", "libroctx64.so.4",
                                 "libroctx64.so"}) {
            if (void *h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) {
                api.push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                api.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (api.push && api.pop) return;
                api = Roctx{};
            }
        }
    });
    return api;
}
}  // namespace

TraceRange::TraceRange(const char *phase, const std::string &field)
{
    const Roctx &r = roctx();
    if (!r.push) return;
    const std::string name = std::string("ogl:") + phase + ":" + field;
    r.push(name.c_str());
    pushed_ = true;
}

TraceRange::~TraceRange()
{
    if (pushed_) roctx().pop();
}

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

// OGL_SEGV_TRACE=1: print a native backtrace on SIGSEGV / SIGABRT (the GPU boxes have no debugger)
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
namespace {
void segv_trace(int sig)
{
    void *frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "libogl_amd: fatal signal, native backtrace:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
struct InstallSegvTrace {
    InstallSegvTrace()
    {
        const char *e = getenv("OGL_SEGV_TRACE");
        if (e && *e == '1') {
            signal(SIGSEGV, segv_trace);
            signal(SIGABRT, segv_trace);
        }
    }
} install_segv_trace;
}  // namespace
