// ipx_fields.h -- packed 16-bit "field pair" helpers shared by the convolution kernels (device only).
//
// A lane owns 16 consecutive bytes of an image row plus 4*HD halo bytes on either side, held as dwords (Win).
// fld(W, p) is the pair (byte p, byte p+2) zero-extended to 2 x u16 in one VGPR (one v_perm_b32); the 8 "slots" of a
// lane, s = 2k + parity, cover bytes (4k+parity, 4k+parity+2) of its 16 output bytes.  Horizontal filter taps are
// then packed 16-bit multiply-adds on such pairs (v_pk_mad_u16); results that need more than 16 bits are
// accumulated per half with v_mad_u32_u16 (op_sel picks the half, no unpacking) and the result bytes are picked
// with v_perm_b32.
#pragma once
#include "ipx_common.h"

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef short ss2 __attribute__((ext_vector_type(2)));

#ifndef IPX_ABLATE
#define IPX_ABLATE 0
#endif
#ifndef IPX_CONV_NT_LOADS
#define IPX_CONV_NT_LOADS 0
#endif

template <int C, int R>
struct Win {                            // bytes [b0 - 4*HD, b0 + 16 + 4*HD) of one row, as dwords
    static constexpr int HD = (R * C + 3) / 4;
    static constexpr int ND = 4 + 2 * HD;
    uint32_t w[ND];
};

// b0 = first of the lane's 16 row bytes, L = row bytes.  The two lanes at the ends of a row have no neighbours to take
// halo bytes from: their halo loads are redirected to bytes inside the row (never before the row's first or behind its
// last byte: the row may be the buffer's first or last) and fix_edges() then builds the halo by REFLECT_101.
// (Tried and dropped, round 2: halo dwords from the neighbouring lanes' registers by DPP wave shifts, with memory
// loads only for the lanes at a wave's ends: 2 % faster for 3 and 5 taps on BGR, 30 % slower for 7 taps -- the loads
// inside divergent branches cost the compiler its exact vmcnt bookkeeping; and nt main loads: 8 % slower per step.)
template <int C, int R>
__device__ __forceinline__ void load_win(Win<C, R>& W, const uint8_t* rp, int b0, int L)
{
    constexpr int HD = Win<C, R>::HD;
#if IPX_CONV_NT_LOADS
    const u32x4 c = ld16_nt(rp + b0);
    __builtin_memcpy(&W.w[HD], &c, 16);
#else
    __builtin_memcpy(&W.w[HD], rp + b0, 16);
#endif
#if IPX_ABLATE & 1          // dev ablation: no halo loads (results are wrong; timing only)
#pragma unroll
    for (int i = 0; i < HD; ++i) { W.w[i] = W.w[HD + i]; W.w[HD + 4 + i] = W.w[HD + 3 - i]; }
#else
    const int lo = b0 >= 4 * HD ? b0 - 4 * HD : b0;
    const int hi = b0 + 16 + 4 * HD <= L ? b0 + 16 : b0 + 16 - 4 * HD;
    __builtin_memcpy(&W.w[0], rp + lo, 4 * HD);
    __builtin_memcpy(&W.w[HD + 4], rp + hi, 4 * HD);
#endif
}

// Halo bytes of the lane that starts a row (b0 == 0) / ends it (b0 + 16 == L), mirrored about the first / last PIXEL
// (cv::BORDER_REFLECT_101, channels interleaved): position -m is pixel -ceil(m / C) -> pixel +ceil(m / C), all of which
// lie inside the lane's own 16 bytes because (R + 1) * C <= 16.  Only the R * C bytes the taps reach are produced.
// Divergent, but only in the two waves of a row that hold these lanes.
template <int C, int R>
__device__ __forceinline__ void fix_edges(Win<C, R>& W, bool first, bool last)
{
    constexpr int HD = Win<C, R>::HD;
    static_assert((R + 1) * C <= 16, "mirrored pixels must lie inside the lane's own 16 bytes");
    if (first) {
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            uint32_t out = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int m = 4 * (HD - d) - k;                    // this byte is row position -m
                const int px = (m + C - 1) / C, ch = px * C - m;   // pixel -px, channel ch
                if (px <= R) {
                    const int sb = px * C + ch;                    // its mirror image: pixel +px
                    out |= ((W.w[HD + (sb >> 2)] >> (8 * (sb & 3))) & 255u) << (8 * k);
                }
            }
            W.w[d] = out;
        }
    }
    if (last) {
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            uint32_t out = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = 4 * d + k;                           // this byte is row position L + t: pixel w + t / C
                const int j = t / C, ch = t - j * C;
                if (j < R) {
                    const int sb = 16 - (2 + j) * C + ch;          // pixel w - 2 - j, relative to b0 = L - 16
                    out |= ((W.w[HD + (sb >> 2)] >> (8 * (sb & 3))) & 255u) << (8 * k);
                }
            }
            W.w[HD + 4 + d] = out;
        }
    }
}

// Field pair (byte p, byte p+2) of the window, zero-extended to 2 x u16; p is relative to b0 and static.
template <int C, int R>
__device__ __forceinline__ us2 fld(const Win<C, R>& W, int p)
{
    const int q = p + 4 * Win<C, R>::HD;
    const int d = q >> 2, j = q & 3;
    const uint32_t lo = W.w[d];
    uint32_t v;
    if (j == 0) {
        v = lo & 0x00ff00ffu;
    } else {
        const uint32_t hi = j >= 2 ? W.w[d + 1] : lo;          // byte p+2 lives in the next dword
        v = __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (uint32_t)j | ((uint32_t)(j + 2) << 16));
    }
    return __builtin_bit_cast(us2, v);
}

// The 8 field pairs of a lane's 16 output bytes: slot s = 2*k + parity covers bytes (4k+parity, 4k+parity+2).
__device__ __forceinline__ constexpr int slot_p(int s) { return 4 * (s >> 1) + (s & 1); }

__device__ __forceinline__ uint32_t pack_fields(us2 even, us2 odd)      // fields hold 0..255
{
    return __builtin_bit_cast(uint32_t, even) | (__builtin_bit_cast(uint32_t, odd) << 8);
}

// D = lo16(p) * c + acc   /   D = hi16(p) * c + acc      (c in an SGPR)
__device__ __forceinline__ uint32_t mad_lo16(uint32_t p, uint32_t c, uint32_t acc)
{
    uint32_t d;
    asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(d) : "v"(p), "s"(c), "v"(acc));
    return d;
}
__device__ __forceinline__ uint32_t mad_hi16(uint32_t p, uint32_t c, uint32_t acc)
{
    uint32_t d;
    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(p), "s"(c), "v"(acc));
    return d;
}

// Output dword from four 32-bit accumulators whose result byte sits in bits 16..23:
// eL -> byte 0, oL -> byte 1, eH -> byte 2, oH -> byte 3.
__device__ __forceinline__ uint32_t pack_acc(uint32_t eL, uint32_t oL, uint32_t eH, uint32_t oH)
{
    return __builtin_amdgcn_perm(oL, eL, 0x0c0c0602u) | __builtin_amdgcn_perm(oH, eH, 0x06020c0cu);
}

