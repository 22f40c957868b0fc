// mgx_ext64.hip — the extension half of the aligner's wave program with all 64 lanes of a wavefront on ONE read (wave.hpp), for
// batches that are spread over the wavefronts anyway (fewer reads than resident wavefronts: long-read batches, single queries;
// mgx.hip, launch_groups).  The 8-lane groups of mgx_grp.hip exist to keep 8 reads per wavefront in flight; a read that has a
// wavefront to itself computes its columns 64 cells at a time instead (a 256-cell chain window, one pass over any band).  Same
// sources, own namespace; carries the alternative paths and the CanonicalDBG branches, so one build serves every graph mode.
#include <hip/hip_runtime.h>

#define mgx mgx_ext64
#define MGX_WITH_PRIMARY 1
#ifndef MGX_MAX_ALT
#define MGX_MAX_ALT 4
#endif
#define MGX_ALIGN_WAVES_PER_SIMD 2
#include "wave.hpp"
#include "seed_kernel.hpp"

using namespace mgx;

extern "C" int mgx_launch_ext64(const void *params, uint32_t blocks, uint32_t lds_bytes, void *stream) {
    const AlignParams &P = *static_cast<const AlignParams *>(params);
    k_align<PH_EXTEND><<<blocks, 64, lds_bytes, (hipStream_t)stream>>>(P, lds_bytes);
    return (int)hipGetLastError();
}
extern "C" unsigned mgx_ext64_static_lds(void) { return (unsigned)(sizeof(Wave) + sizeof(SdustScratch) + 6 * 128); }
extern "C" int mgx_ext64_waves_per_simd(void) { return MGX_ALIGN_WAVES_PER_SIMD; }
