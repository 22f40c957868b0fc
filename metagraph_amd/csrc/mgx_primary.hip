// mgx_primary.hip — the seeding kernel with the CanonicalDBG (PRIMARY graph) branches of align_core.hpp compiled in
// (reverse-complement sub-k seeds, wrapper terminus bits; canon_graph.hpp).  A separate instantiation in its own namespace,
// so that the seeding kernel of every other graph (mgx.hip) stays the code it was.  The extension half for PRIMARY graphs is
// the MGX_WITH_PRIMARY build of mgx_grp.hip.
#include <hip/hip_runtime.h>

#define mgx mgx_primary
#define MGX_WITH_PRIMARY 1
#define MGX_NO_EXTEND 1
#include "wave.hpp"
#include "seed_kernel.hpp"

using namespace mgx;

extern "C" int mgx_launch_seed_primary(const void *params, uint32_t blocks, uint32_t lds_bytes, int wps8, void *stream) {
    const AlignParams &P = *static_cast<const AlignParams *>(params);
    if (wps8) k_align<PH_SEED, MGX_SEED_WPS><<<blocks, 64, lds_bytes, (hipStream_t)stream>>>(P, lds_bytes);
    else k_align<PH_SEED><<<blocks, 64, lds_bytes, (hipStream_t)stream>>>(P, lds_bytes);
    return (int)hipGetLastError();
}
