// GPU check of the sub-wave-group primitives of metagraph_amd/csrc/wave_group.hpp (8 lanes per group): the DPP butterfly
// reductions, the masked-OR broadcast, prefix max, shifts and ballots against plain host loops, with the groups of the
// wavefront holding different data and with some groups masked off (divergent groups must not disturb each other).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define MGX_GROUP 8
#include "../../metagraph_amd/csrc/wave_group.hpp"
using namespace mgx;
__global__ void k(const int *in, const int *src, unsigned active_groups, int *out) {
    LV<int32_t> x; x.v = in[threadIdx.x];
    const int g = group_id();
    int *o = out + threadIdx.x;
    for (int t = 0; t < 9; ++t) o[64 * t] = -12345;
    if (!((active_groups >> g) & 1u)) return;               // whole groups leave: the others carry on
    o[0] = wave_max(x);
    o[64] = wave_min(x);
    o[128] = wave_bcast(x, src[g]);
    o[192] = wave_prefix_max(x).v;
    o[256] = wave_shift_up1(x, -7).v;
    o[320] = wave_shift_down(x, src[g] & 3, -9).v;
    o[384] = wave_sum(x);
    LV<bool> p; p.v = (x.v & 1) != 0;
    o[448] = (int)wave_ballot(p);
    LV<uint64_t> y; y.v = ((uint64_t)(uint32_t)x.v << 32) | (uint32_t)(x.v * 7 + 3);
    const uint64_t b = wave_bcast(y, src[g]);
    o[512] = (int)(uint32_t)(b >> 32) ^ (int)(uint32_t)b;
}
int main() {
    int h[64], hs[8], *d, *ds, *o;
    hipMalloc(&d, 256); hipMalloc(&ds, 32); hipMalloc(&o, 9 * 256);
    int bad = 0;
    srand(7);
    for (int t = 0; t < 200; ++t) {
        for (int i = 0; i < 64; ++i) h[i] = (rand() % 2001) - 1000 + (i % 13 == 5 && t % 3 == 0 ? INT32_MIN + 100 : 0) + (t % 7 == 2 && i % 9 == 1 ? INT32_MAX - 2000 : 0);
        for (int g = 0; g < 8; ++g) hs[g] = rand() % 8;
        const unsigned act = t % 4 == 0 ? 0xFFu : (unsigned)(rand() & 0xFF);
        hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
        hipMemcpy(ds, hs, 32, hipMemcpyHostToDevice);
        k<<<1, 64>>>(d, ds, act, o);
        int r[9 * 64];
        hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
        for (int g = 0; g < 8; ++g) {
            const int *v = h + 8 * g;
            if (!((act >> g) & 1u)) { for (int l = 0; l < 8; ++l) for (int q = 0; q < 9; ++q) if (r[64 * q + 8 * g + l] != -12345) ++bad; continue; }
            int mx = INT32_MIN, mn = INT32_MAX, sum = 0, bal = 0, pm = INT32_MIN;
            for (int l = 0; l < 8; ++l) { mx = v[l] > mx ? v[l] : mx; mn = v[l] < mn ? v[l] : mn; sum += v[l]; bal |= (v[l] & 1) << l; }
            for (int l = 0; l < 8; ++l) {
                pm = v[l] > pm ? v[l] : pm;
                const int i = 8 * g + l;
                if (r[i] != mx) ++bad;
                if (r[64 + i] != mn) ++bad;
                if (r[128 + i] != v[hs[g]]) ++bad;
                if (r[192 + i] != pm) ++bad;
                if (r[256 + i] != (l ? v[l - 1] : -7)) ++bad;
                const int sh = hs[g] & 3;
                if (r[320 + i] != (l + sh < 8 ? v[l + sh] : -9)) ++bad;
                if (r[384 + i] != sum) ++bad;
                if (r[448 + i] != bal) ++bad;
                if (r[512 + i] != (v[hs[g]] ^ (v[hs[g]] * 7 + 3))) ++bad;
            }
        }
    }
    printf("group primitives (8 lanes): %s (%d bad)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
