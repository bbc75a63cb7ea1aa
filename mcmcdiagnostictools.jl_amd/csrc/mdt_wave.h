// mdt_wave.h -- host-side interface of the wave-per-parameter kernel (mdt_wave.hip), used by mdt_hip.hip.
// Internal to libmdt_hip.so: nothing here is part of the C-ABI (include/mdt_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "mdt_kp.hpp"

namespace mdtw {

// limits of the path (everything else is served by the workgroup-per-parameter kernels of mdt_device.hpp)
constexpr int WV_MAXN = 8192;     // draws * chains
constexpr int WV_MAXM = 16;       // chains * split_chains
constexpr int WV_NITER = 1024;    // draws / split_chains (above WV_NITER_SHORT the per-wave LDS block is the longer one: fewer waves per CU)
constexpr int WV_NITER_SHORT = 640;
constexpr int WV_MAXLAG = 255;    // effective maxlag

// per-launch geometry, fixed by the slice size and the element type
struct Geo {
    int perwave;    // bytes of LDS per wave: the block of short split chains or the longer one
    int waves;          // waves per workgroup (one workgroup per CU)
    int tab_bytes;      // LDS bytes of the normal-score table shared by the workgroup
    int lds_bytes;      // dynamic LDS of the launch
    long long npad;     // elements of per-wave scratch
};
// Diagnostics of a request (read by the caller behind the MDT_TEST_HOOKS gate of mdt_hip.hip and handed over BY VALUE: the unit
// keeps no mutable state, so contexts on several threads -- the workers of a multi-GPU context, stream contexts -- never share any).
struct Hooks {
    int waves = 0;      // at most this many waves per workgroup (0 = what the LDS allows; MDT_WAVE_WAVES)
    int tab_mode = 0;   // 0: the rule of geometry(), 1: score table always in LDS, 2: never (MDT_WAVE_TAB)
    int stop = 0;       // parameters end after phase k (0 = never; MDT_WAVE_STOP)
    int osonly = 1;     // ess(kind = tail) alone ranks only the rounds that hold its quantiles (0: the full ranking; MDT_WAVE_OSONLY)
    unsigned* dbg = nullptr; // progress word per wave in host-pinned memory; honoured by -DWV_DEBUG / -DWV_PROF builds only (MDT_WAVE_DBG)
};
Geo geometry(long long n, long long niter, int f32, const Hooks& hk = Hooks(), int raw = 0); // raw: a raw-draw request (no score table)
// bytes of per-wave scratch in global memory for `waves_total` waves
size_t scratch_bytes(const Geo& g, long long waves_total);
// sizeof(KP) as this unit was compiled: mdt_ctx_create compares it with its own (a variant build that links a stale partner unit
// -- tools/build_variant.sh -- would otherwise read a parameter block of another layout)
size_t kp_bytes();
int prepare_device(); // once per device, under its hipSetDevice: 0 = ok
// Queue the kernel on `st`: workgroups x waves tickets over kp->P parameters.  `kp_slot` is the device copy of the parameter
// block (kp_setup_kernel wrote it on the same stream).  Parameters the path cannot serve (a non-finite value, a point mass
// larger than a round, too many ties among the folded values) are appended to fb_list, their count to *fb_count.
void launch(hipStream_t st, const mdtc::KP* kp_slot, int f32, const Geo& g, int workgroups, void* scratch,
            unsigned* fb_list, int* fb_count, const Hooks& hk = Hooks());

} // namespace mdtw
