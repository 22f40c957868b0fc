// mgx_lab64.hip — the extension half of the aligner's wave program with the label-aware extender compiled in
// (LabeledAligner<>: A/aligner_labeled.{hpp,cpp}; label_sets.hpp / label_driver.hpp), one read per wavefront, all 64 lanes
// on its columns — the shape of mgx_ext64.hip.  Launched for the batches of an aligner made by mgx_labeled_aligner_create;
// every other batch runs the kernels without the label hooks.  Same sources, own namespace.
#include <hip/hip_runtime.h>

#define mgx mgx_lab64
#define MGX_WITH_LABELS 1
#define MGX_WITH_PRIMARY 1
#ifndef MGX_MAX_ALT
#define MGX_MAX_ALT 4
#endif
#define MGX_ALIGN_WAVES_PER_SIMD 2
#include "wave.hpp"
#include "seed_kernel.hpp"

using namespace mgx;

extern "C" int mgx_launch_lab64(const void *params, uint32_t blocks, uint32_t lds_bytes, void *stream) {
    const AlignParams &P = *static_cast<const AlignParams *>(params);
    k_align<PH_EXTEND><<<blocks, 64, lds_bytes, (hipStream_t)stream>>>(P, lds_bytes);
    return (int)hipGetLastError();
}
extern "C" unsigned mgx_lab64_static_lds(void) { return (unsigned)(sizeof(Wave) + sizeof(SdustScratch) + 6 * 128); }
extern "C" int mgx_lab64_waves_per_simd(void) { return MGX_ALIGN_WAVES_PER_SIMD; }
extern "C" int mgx_lab64_max_alt(void) { return MGX_MAX_ALT; }
