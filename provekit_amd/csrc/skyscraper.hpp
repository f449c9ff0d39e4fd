// skyscraper.hpp -- Skyscraper constants shared by host and device code: the round
// constants (skyscraper/core/src/constants.rs:30-49) and the byte S-box (bar.rs:40-67).  The compression itself
// (reference.rs:41-98 v2, v1.rs:19-32 v1; structure of generic.rs:77-102) is in skyscraper29.hpp.
// Values are canonical integers (NOT Montgomery): the hash squares with a Montgomery product, which is exactly
// sq(x) = x^2 * 2^-256 mod p (reference.rs:22-26: SIGMA_INV == 2^-256).
#pragma once
#include "fe.hpp"

namespace pk {

// skyscraper/core/src/constants.rs:30-49, as 8 x u32 little-endian limbs
__host__ __device__ __forceinline__ constexpr u32 rc_limb(int rc, int i) {
    constexpr u64 RC[18][4] = {
        {0x0000000000000000ULL, 0x0000000000000000ULL, 0x0000000000000000ULL, 0x0000000000000000ULL},
        {0x903c4324270bd744ULL, 0x873125f708a7d269ULL, 0x081dd27906c83855ULL, 0x276b1823ea6d7667ULL},
        {0x7ac8edbb4b378d71ULL, 0xe29d79f3d99e2cb7ULL, 0x751417914c1a5a18ULL, 0x0cf02bd758a484a6ULL},
        {0xfa7adc6769e5bc36ULL, 0x1c3f8e297cca387dULL, 0x0eb7730d63481db0ULL, 0x25b0e03f18ede544ULL},
        {0x57847e652f03cfb7ULL, 0x33440b9668873404ULL, 0x955a32e849af80bcULL, 0x002882fcbe14ae70ULL},
        {0x979231396257d4d7ULL, 0x29989c3e1b37d3c1ULL, 0x12ef02b47f1277baULL, 0x039ad8571e2b7a9cULL},
        {0xb5b48465abbb7887ULL, 0xa72a6bc5e6ba2d2bULL, 0x4cd48043712f7b29ULL, 0x1142d5410fc1fc1aULL},
        {0x7ab2c156059075d3ULL, 0x17cb3594047999b2ULL, 0x44f2c93598f289f7ULL, 0x1d78439f69bc0becULL},
        {0x05d7a965138b8edbULL, 0x36ef35a3d55c48b1ULL, 0x8ddfb8a1ac6f1628ULL, 0x258588a508f4ff82ULL},
        {0x1596fb9afccb49e9ULL, 0x9a7367d69a09a95bULL, 0x9bc43f6984e4c157ULL, 0x13087879d2f514feULL},
        {0x295ccd233b4109faULL, 0xe1d72f89ed868012ULL, 0x2e9e1eea4bc88a8eULL, 0x17dadee898c45232ULL},
        {0x9a8590b4aa1f486fULL, 0xb75834b430e9130eULL, 0xb8e90b1034d5de31ULL, 0x295c6d1546e7f4a6ULL},
        {0x850adcb74c6eb892ULL, 0x07699ef305b92fc3ULL, 0x4ef96a2ba1720f2dULL, 0x1288ca0e1d3ed446ULL},
        {0x01960f9349d1b5eeULL, 0x8ccad30769371c69ULL, 0xe5c81e8991c98662ULL, 0x17563b4d1ae023f3ULL},
        {0x6ba01e9476b32917ULL, 0xa1cb0a3add977bc9ULL, 0x86815a945815f030ULL, 0x2869043be91a1eeaULL},
        {0x81776c885511d976ULL, 0x7475d34f47f414e7ULL, 0x5d090056095d96cfULL, 0x14941f0aff59e79aULL},
        {0xbc40b4fd8fc8c034ULL, 0xbb7142c3cce4fd48ULL, 0x318356758a39005aULL, 0x1ce337a190f4379fULL},
        {0x0000000000000000ULL, 0x0000000000000000ULL, 0x0000000000000000ULL, 0x0000000000000000ULL},
    };
    return (u32)(RC[rc][i >> 1] >> ((i & 1) * 32));
}

// byte-wise S-box on 4 packed bytes (skyscraper/core/src/bar.rs:40-42,63-67): per byte
//   sbox(v) = rotl1(v ^ (rotl1(~v) & rotl2(v) & rotl3(v))).
// Rotation commutes with the bitwise operations, so with W = ~v & rotl1(v) & rotl2(v) this is rotl1(v) ^ rotl2(W): three
// byte-lane rotations instead of four, each two shifts and one bit-select.
__host__ __device__ __forceinline__ u32 rotl_bytes(u32 v, int s) {
    const u32 keep = (u32)((0xffu << s) & 0xffu) * 0x01010101u;  // bits that receive the left-shifted part
    return ((v << s) & keep) | ((v >> (8 - s)) & ~keep);
}
__host__ __device__ __forceinline__ u32 sbox4(u32 v) {
    const u32 r1 = rotl_bytes(v, 1), r2 = rotl_bytes(v, 2);
    const u32 w = ~v & r1 & r2;
    return r1 ^ rotl_bytes(w, 2);
}

}  // namespace pk
