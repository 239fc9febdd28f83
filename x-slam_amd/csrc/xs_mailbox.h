// xs_mailbox.h — device side of the pose mailbox shared by k_icp<POSE_POSTED> (xs_icp.hip) and the posted integrate kernel
// (xs_tsdf.hip).  xs_icp_post_pose (host) writes it, a kernel that was enqueued before its pose existed polls it: two 64-byte
// lines of 16 words, each line starting with the sequence number:   line 0 = {seq, cmd, f[0..13]}   line 1 = {seq, 0, f[14..23], pad}
// with f = 18 floats of a complex 3x3 followed by the 6 of a complex 3-vector; cmd 0 = run, 1 = abandon the launch.  The mailbox lives
// in device memory the CPU reaches through the large BAR (xs_icp_mailbox_alloc), so polling stays off the PCIe link.
#pragma once
#include <hip/hip_runtime.h>

namespace xs {
enum { MAILBOX_WORDS = 32, MAILBOX_MAX_POLLS = 400000 };  // ~2 us per poll: gives up after about a second

// One wave of the workgroup (the caller passes its lane) polls until both lines carry `seq` — or a later number: the host has
// moved past this launch (one abandon command releases every launch in the queue) — and leaves the 32 words in s_mail, with
// s_mail[1] = 0 run (payload valid), 1 abandon, 2 gave up after MAILBOX_MAX_POLLS / the one-post-per-launch contract was broken.
// The caller follows with a workgroup barrier.
__device__ __forceinline__ void mailbox_wait(const unsigned *mailbox, unsigned seq, unsigned *s_mail, int lane) {
    unsigned v = 0, cmd_override = 0;
    for (int polls = 0;; ++polls) {
        v = __hip_atomic_load(mailbox + (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned s0 = __builtin_amdgcn_readlane(v, 0), s1 = __builtin_amdgcn_readlane(v, 16);
        if (s0 == s1 && (int)(s0 - seq) >= 0) break;
        if (polls >= MAILBOX_MAX_POLLS) { cmd_override = 2; break; }
        __builtin_amdgcn_s_sleep(8);
    }
    // The load that saw both sequence words is not taken as the payload: sixteen lanes reading one line are one request in
    // practice, but nothing promises that its sectors are read at one instant, and a line caught between the host's payload
    // stores and its sequence store would hand over a mixed pose without any error.  The host orders payload -> store fence ->
    // sequence words -> store fence (xs_icp_post_pose), so a load ISSUED after the sequence words were seen returns the
    // complete payload: read the 32 words once more (the exit test above consumed v, i.e. the first load has returned before
    // this one is issued; both are system-scope and bypass the caches).  The sequence words are checked again on the way.
    if (!cmd_override) {
        v = __hip_atomic_load(mailbox + (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned s0 = __builtin_amdgcn_readlane(v, 0), s1 = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 1);
        if (s0 == seq && s1 == seq) { /* this launch's post: pose or command as posted */ }
        else if (s0 == s1 && (int)(s0 - seq) > 0 && c == 1) cmd_override = 1;   // abandon, addressed to a later launch: leave too
        else cmd_override = 2;   // the host broke the one-post-per-launch contract
    }
    if (lane < MAILBOX_WORDS) s_mail[lane] = lane == 1 && cmd_override ? cmd_override : v;
}
// payload float i (0..23) of a mailbox image in LDS, as a wave-uniform value (scalar register)
__device__ __forceinline__ float mailbox_float(const unsigned *s_mail, int i) {
    return __int_as_float(__builtin_amdgcn_readfirstlane((int)s_mail[i < 14 ? 2 + i : 18 + (i - 14)]));
}
}  // namespace xs

// ---- host side: how the CPU writes a mailbox line ---------------------------------------------------------------------------------
// The mailbox is device memory behind the PCIe BAR: write-combining on the CPU side.  Ordinary stores wait in the core's write-combining
// buffers until something drains them, and may pass each other until a store fence — so a post was payload, sfence, sequence words, sfence,
// each fence a round of draining while the kernel waits (profiles/r06_ab_gn_post_fences.txt: a quarter of a microsecond apiece).  Where the CPU
// has MOVDIR64B (cpuid 7.0 ecx bit 28: Sapphire Rapids, Zen 5) a whole 64-byte line — its sequence word and its payload together — goes out as
// ONE write transaction that never sits in a buffer: no fence between payload and sequence word (they cannot be seen apart), none behind it.
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <cpuid.h>
#include <immintrin.h>
namespace xs {
__attribute__((target("movdir64b"))) static inline void mailbox_direct_store_64(void *dst64, const void *src64) { _movdir64b(dst64, src64); }
static inline bool mailbox_cpu_has_direct_store() {
    static const bool has = [] { unsigned a = 0, b = 0, c = 0, d = 0; return __get_cpuid_count(7, 0, &a, &b, &c, &d) && ((c >> 28) & 1u); }();
    return has;
}
static inline void mailbox_store_fence() { __builtin_ia32_sfence(); }
}  // namespace xs
#else   // (the device pass parses host functions too: it sees these)
namespace xs {
static inline void mailbox_direct_store_64(void *, const void *) {}
static inline bool mailbox_cpu_has_direct_store() { return false; }
static inline void mailbox_store_fence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
}  // namespace xs
#endif
#include <cstring>
namespace xs {
// the two lines of one mailbox as they lie in memory: line 0 = {seq, cmd, f[0..13]}, line 1 = {seq, 0, f[14..23], pad}
static inline void mailbox_image(unsigned img[MAILBOX_WORDS], const float *R18, const float *t6, unsigned seq, int cmd) {
    unsigned f[24] = {0};
    if (R18) std::memcpy(f, R18, 18 * sizeof(float));
    if (t6) std::memcpy(f + 18, t6, 6 * sizeof(float));
    std::memset(img, 0, MAILBOX_WORDS * sizeof(unsigned));
    img[0] = seq; img[1] = (unsigned)cmd;
    for (int i = 0; i < 14; ++i) img[2 + i] = f[i];
    img[16] = seq;
    for (int i = 0; i < 10; ++i) img[18 + i] = f[14 + i];
}
}  // namespace xs
