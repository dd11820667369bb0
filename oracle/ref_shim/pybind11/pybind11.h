// Stand-in for pybind11 (TEST INFRASTRUCTURE, see ../Eigen/Dense): the reference's PYBIND11_MODULE
// block only registers its functions with Python; here it compiles to an unused function and the
// C wrapper (oracle/ref_cfar_wrap.cpp) calls ca / soca / goca / os / *2 directly.
#pragma once
#include <cmath>
#include <cstdlib>
#include <utility>
namespace pybind11 {
struct module_stub {
    template <typename F>
    module_stub &def(const char *, F)
    {
        return *this;
    }
};
} // namespace pybind11
#define PYBIND11_MODULE(name, var) static void pybind11_stub_##name(pybind11::module_stub &var)
