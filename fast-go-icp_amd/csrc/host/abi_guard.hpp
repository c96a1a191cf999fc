// Exception barrier of the C ABI (include/fgoicp_amd.h: "every call returns an fgoicp_status and never throws").
#pragma once
#include <exception>
#include <new>
#include <string>

#include "../../../include/fgoicp_amd.h"

namespace fgoicp {
void set_error(const std::string& s);  // csrc/device/ctx.hip

// The C ABI never throws (include/fgoicp_amd.h): the entry points that allocate on the host (contexts, solvers, whole runs) run their
// bodies behind this barrier — std::bad_alloc becomes FGOICP_ERR_OOM, anything else FGOICP_ERR_HIP with the exception's text.
template <class F>
int abi_guard(const char* what, F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        try { set_error(std::string(what) + ": out of host memory"); } catch (...) {}
        return FGOICP_ERR_OOM;
    } catch (const std::exception& e) {
        try { set_error(std::string(what) + ": " + e.what()); } catch (...) {}
        return FGOICP_ERR_HIP;
    } catch (...) {
        try { set_error(std::string(what) + ": unknown exception"); } catch (...) {}
        return FGOICP_ERR_HIP;
    }
}

}  // namespace fgoicp
