// The library's fifth_power (Schlick's (1 - cos)^5, materials.py:369) compiled for the host, for
// tests/test_fifth_power.py:
//     hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -std=c++17 -fPIC -shared tests/native/fifth_power_host.hip -o <so>
#include "../../raytracing-one-weekend_amd/csrc/wrt_device.hpp"

extern "C" void fifth_power_many(const double* x, double* y, long long n) {
    for (long long i = 0; i < n; ++i) y[i] = fifth_power(x[i]);
}
