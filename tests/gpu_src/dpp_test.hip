#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../metagraph_amd/csrc/wave.hpp"
using namespace mgx;
__global__ void k(const int* in, int* pm, int* mx, int* mn, int* sh, int* bc) {
    LV<int32_t> x; x.v = in[threadIdx.x];
    pm[threadIdx.x] = wave_prefix_max(x).v;
    mx[threadIdx.x] = wave_max(x);
    mn[threadIdx.x] = wave_min(x);
    sh[threadIdx.x] = wave_shift_up1(x, -7).v;
    bc[threadIdx.x] = wave_bcast(x, 37);
}
int main() {
    int h[64], *d, *o; hipMalloc(&d, 256); hipMalloc(&o, 5*256);
    int bad = 0;
    for (int t = 0; t < 50; ++t) {
        for (int i = 0; i < 64; ++i) h[i] = (rand() % 2001) - 1000 + (i == 13 && t % 3 == 0 ? INT32_MIN + 100 : 0);
        hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
        k<<<1, 64>>>(d, o, o + 64, o + 128, o + 192, o + 256);
        int r[320]; hipMemcpy(r, o, 1280, hipMemcpyDeviceToHost);
        int m = INT32_MIN, mn = INT32_MAX;
        for (int i = 0; i < 64; ++i) { m = h[i] > m ? h[i] : m; mn = h[i] < mn ? h[i] : mn; if (r[i] != m) ++bad; }
        for (int i = 0; i < 64; ++i) { if (r[64+i] != m) ++bad; if (r[128+i] != mn) ++bad; if (r[192+i] != (i ? h[i-1] : -7)) ++bad; if (r[256+i] != h[37]) ++bad; }
    }
    printf("dpp primitives: %s (%d bad)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
