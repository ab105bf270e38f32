// kvm_seqadd.h -- one lane adds fp64 values from LDS strictly in order (the exact pass, kvm_exact.hip; the inline exact
// evaluation of kvm_ed_dot.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- the sequential chains -----------------------------------------------------------------------------------------
// One lane adds values from LDS strictly in order.  A dependent v_add_f64 issues every ~10 cycles
// (profiles/micro/chain_bench.hip), so the loop is at its floor only if no LDS latency is ever exposed: two register
// sets alternate, the reads of one are in flight while the other is added.  hipcc would not keep that shape -- it
// sank the prefetch next to its use, or renamed the sets and waited on the copies (26-31 cycles per step measured with
// profiles/micro/replay_timing.hip) -- so the reads and their waits are written as inline asm: the compiler sees plain
// register operands, the s_waitcnt that makes a set valid carries the set and the running sums as operands (nothing
// that uses them can move above it), and lgkmcnt(8) means "all but the 8 youngest reads", i.e. the other set's.
typedef double ex_d2 __attribute__((ext_vector_type(2)));
#define EX_LDS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define EX_READ8(set, addr, stride)                                                                                      \
    do {                                                                                                                 \
        if (stride == 16) {                                                                                              \
            EX_LDS_READ(set[0], addr, 0); EX_LDS_READ(set[1], addr, 16); EX_LDS_READ(set[2], addr, 32);                  \
            EX_LDS_READ(set[3], addr, 48); EX_LDS_READ(set[4], addr, 64); EX_LDS_READ(set[5], addr, 80);                 \
            EX_LDS_READ(set[6], addr, 96); EX_LDS_READ(set[7], addr, 112);                                               \
        } else {                                                                                                         \
            EX_LDS_READ(set[0], addr, 0); EX_LDS_READ(set[1], addr, 32); EX_LDS_READ(set[2], addr, 64);                  \
            EX_LDS_READ(set[3], addr, 96); EX_LDS_READ(set[4], addr, 128); EX_LDS_READ(set[5], addr, 160);               \
            EX_LDS_READ(set[6], addr, 192); EX_LDS_READ(set[7], addr, 224);                                              \
        }                                                                                                                \
    } while (0)
#define EX_WAIT8(set, s0, s1)                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(8)"                                                                                  \
                 : "+v"(set[0]), "+v"(set[1]), "+v"(set[2]), "+v"(set[3]), "+v"(set[4]), "+v"(set[5]), "+v"(set[6]),     \
                   "+v"(set[7]), "+v"(s0), "+v"(s1)::"memory")
#define EX_WAIT0(set, s0, s1)                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                  \
                 : "+v"(set[0]), "+v"(set[1]), "+v"(set[2]), "+v"(set[3]), "+v"(set[4]), "+v"(set[5]), "+v"(set[6]),     \
                   "+v"(set[7]), "+v"(s0), "+v"(s1)::"memory")
__device__ __forceinline__ unsigned ex_lds_addr(const void *p) { return (unsigned)(uintptr_t)p; }  // low half of a generic LDS address

// lane: acc += buf[0..cnt) strictly in order (16 values per register set).
__device__ __forceinline__ double ex_seq_add(const double *buf, int cnt, double acc) {
    ex_d2 a[8], b[8];
    double dummy = 0.0;
    const int rounds = cnt / 32;  // a round = both sets = 32 values
    if (rounds > 0) {
        const unsigned base = ex_lds_addr(buf);
        EX_READ8(a, base, 16);
        for (int r = 0; r < rounds; r++) {
            const unsigned ab = base + (unsigned)(r * 256 + 128);
            const unsigned an = base + (unsigned)((r + 1 < rounds ? r + 1 : 0) * 256);  // (last: re-reads the head, unused)
            EX_READ8(b, ab, 16);
            EX_WAIT8(a, acc, dummy);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                acc += a[u].x;
                acc += a[u].y;
            }
            EX_READ8(a, an, 16);
            EX_WAIT8(b, acc, dummy);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                acc += b[u].x;
                acc += b[u].y;
            }
        }
        EX_WAIT0(a, acc, dummy);  // drain the last (unused) prefetch before the compiler's own LDS code
    }
    for (int t = rounds * 32; t < cnt; t++) acc += buf[t];
    return acc;
}

