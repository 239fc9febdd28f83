// xs_mailbox.h — the pose mailbox shared by k_icp<POSE_POSTED> (xs_icp.hip), the posted integrate kernel and the posted Gauss-Newton pass
// (xs_tsdf.hip).  xs_icp_post_pose (host) writes it, a kernel that was enqueued before its pose existed polls it.  128 bytes = four 32-byte
// SECTORS of eight words, EVERY sector starting with the sequence number:
//     sector 0 = {seq, cmd, f[0..5]}   sector 1 = {seq, f[6..12]}   sector 2 = {seq, 0, f[13..18]}   sector 3 = {seq, f[19..23], 0, 0}
// with f = the 18 floats of a complex 3x3 followed by the 6 of a complex 3-vector; cmd 0 = run, 1 = abandon the launch.  A 32-byte sector is the
// smallest piece of memory a load is served from: a load that finds the launch's number in all four sectors holds the complete payload — no
// second read (round 6; until then the number stood once per 64-byte line and the payload was read again after both had been seen: 0.35 us per
// launch, 1.5 % of a frame, profiles/r06_ab_mailbox_reread.txt).  The mailbox lives in device memory the CPU reaches through the large BAR
// (xs_icp_mailbox_alloc), so polling stays off the PCIe link.
#pragma once
#include <hip/hip_runtime.h>

namespace xs {
enum { MAILBOX_WORDS = 32, MAILBOX_MAX_POLLS = 400000 };  // ~2 us per poll: gives up after about a second
// word of payload float i (0 .. 23), and payload float of word w (-1: a sequence word, cmd, or padding)
__host__ __device__ constexpr int mailbox_word_of(int i) { return i < 6 ? 2 + i : (i < 13 ? 9 + (i - 6) : (i < 19 ? 18 + (i - 13) : 25 + (i - 19))); }
__host__ __device__ constexpr int mailbox_float_of(int w) {
    return w >= 2 && w < 8 ? w - 2 : (w >= 9 && w < 16 ? w - 3 : (w >= 18 && w < 24 ? w - 5 : (w >= 25 && w < 30 ? w - 6 : -1)));
}

// One wave of the workgroup (the caller passes its lane) polls until all four sectors carry `seq` — or a later number: the host has
// moved past this launch (one abandon command releases every launch in the queue) — and leaves the 32 words in s_mail, with
// s_mail[1] = 0 run (payload valid), 1 abandon, 2 gave up after MAILBOX_MAX_POLLS / the one-post-per-launch contract was broken.
// The caller follows with a workgroup barrier.
__device__ __forceinline__ void mailbox_wait(const unsigned *mailbox, unsigned seq, unsigned *s_mail, int lane) {
    unsigned v = 0, cmd_override = 0;
    for (int polls = 0;; ++polls) {
        v = __hip_atomic_load(mailbox + (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned s0 = __builtin_amdgcn_readlane(v, 0), s1 = __builtin_amdgcn_readlane(v, 8), s2 = __builtin_amdgcn_readlane(v, 16),
                       s3 = __builtin_amdgcn_readlane(v, 24);
        if (s0 == s1 && s1 == s2 && s2 == s3 && (int)(s0 - seq) >= 0) {
            // every sector of THIS load carries one number: the words are one post's, whole (the host writes a sector's number with or after
            // its payload: xs_icp_post_pose)
            const unsigned c = __builtin_amdgcn_readlane(v, 1);
            if (s0 == seq) { /* this launch's post: pose or command as posted */ }
            else if (c == 1) cmd_override = 1;   // abandon, addressed to a later launch: leave too
            else cmd_override = 2;               // the host broke the one-post-per-launch contract
            break;
        }
        if (polls >= MAILBOX_MAX_POLLS) { cmd_override = 2; break; }
#ifndef XS_MAILBOX_POLL_SLEEP
#define XS_MAILBOX_POLL_SLEEP 8   // (x 64 clocks between two polls of a workgroup; 2 and 0 measured no faster: profiles/r06_ab_mailbox_poll_sleep.txt)
#endif
        if (XS_MAILBOX_POLL_SLEEP > 0) __builtin_amdgcn_s_sleep(XS_MAILBOX_POLL_SLEEP);
    }
    if (lane < MAILBOX_WORDS) s_mail[lane] = lane == 1 && cmd_override ? cmd_override : v;
}
// payload float i (0..23) of a mailbox image in LDS, as a wave-uniform value (scalar register)
__device__ __forceinline__ float mailbox_float(const unsigned *s_mail, int i) {
    return __int_as_float(__builtin_amdgcn_readfirstlane((int)s_mail[mailbox_word_of(i)]));
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
// the four sectors of one mailbox as they lie in memory (layout at the top of this file)
static inline void mailbox_image(unsigned img[MAILBOX_WORDS], const float *R18, const float *t6, unsigned seq, int cmd) {
    unsigned f[24] = {0};
    if (R18) std::memcpy(f, R18, 18 * sizeof(float));
    if (t6) std::memcpy(f + 18, t6, 6 * sizeof(float));
    std::memset(img, 0, MAILBOX_WORDS * sizeof(unsigned));
    for (int i = 0; i < 24; ++i) img[mailbox_word_of(i)] = f[i];
    img[1] = (unsigned)cmd;
    img[0] = img[8] = img[16] = img[24] = seq;
}
// the fenced form of a post (no MOVDIR64B): everything but the four sequence words, a store fence, the four sequence words, a store fence
static inline void mailbox_store_fenced(volatile unsigned *w, const unsigned img[MAILBOX_WORDS]) {
    for (int i = 0; i < MAILBOX_WORDS; ++i) if (i % 8 != 0) w[i] = img[i];
    mailbox_store_fence();
    for (int i = 0; i < MAILBOX_WORDS; i += 8) w[i] = img[i];
    mailbox_store_fence();
}
}  // namespace xs
