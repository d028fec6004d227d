// RFC 1321 MD5 for sketch identities.
// Replaces the reference's dependency `md5 = 0.7.0` as used by
// src/core/src/sketch/minhash.rs:290-307 (md5sum: digest of the decimal ASCII
// of ksize followed by the decimal ASCII of every hash, no separators).
#pragma once
#include <stdint.h>
#include <string.h>
#include <string>

namespace smg {

class Md5 {
  public:
    Md5() : a_(0x67452301u), b_(0xefcdab89u), c_(0x98badcfeu), d_(0x10325476u), total_(0), fill_(0) {}

    void update(const void* data, size_t n) {
        const uint8_t* p = static_cast<const uint8_t*>(data);
        total_ += n;
        if (fill_) {
            const size_t take = n < 64 - fill_ ? n : 64 - fill_;
            memcpy(buf_ + fill_, p, take);
            fill_ += take; p += take; n -= take;
            if (fill_ < 64) return;
            transform(buf_);
            fill_ = 0;
        }
        for (; n >= 64; p += 64, n -= 64) transform(p);
        if (n) { memcpy(buf_, p, n); fill_ = n; }
    }

    void update_decimal(uint64_t v) {
        char tmp[24];
        int i = 24;
        do { tmp[--i] = (char)('0' + v % 10); v /= 10; } while (v);
        update(tmp + i, (size_t)(24 - i));
    }

    std::string hexdigest() {
        const uint64_t bits = total_ * 8;
        static const uint8_t pad[64] = {0x80};
        update(pad, fill_ < 56 ? 56 - fill_ : 120 - fill_);
        uint8_t lenb[8];
        for (int i = 0; i < 8; ++i) lenb[i] = (uint8_t)(bits >> (8 * i));
        update(lenb, 8);
        const uint32_t w[4] = {a_, b_, c_, d_};
        static const char* hexd = "0123456789abcdef";
        std::string out(32, '0');
        for (int i = 0; i < 16; ++i) {
            const uint8_t byte = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
            out[2 * i] = hexd[byte >> 4];
            out[2 * i + 1] = hexd[byte & 15];
        }
        return out;
    }

  private:
    static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }

    void transform(const uint8_t* blk) {
        uint32_t x[16];
        for (int i = 0; i < 16; ++i) memcpy(&x[i], blk + 4 * i, 4);   // little-endian host
        uint32_t a = a_, b = b_, c = c_, d = d_;
#define SMG_MD5_STEP(f, a, b, c, d, xi, t, s) a = b + rol(a + f(b, c, d) + xi + t, s)
#define SMG_F(x, y, z) ((z) ^ ((x) & ((y) ^ (z))))
#define SMG_G(x, y, z) ((y) ^ ((z) & ((x) ^ (y))))
#define SMG_H(x, y, z) ((x) ^ (y) ^ (z))
#define SMG_I(x, y, z) ((y) ^ ((x) | ~(z)))
        SMG_MD5_STEP(SMG_F, a, b, c, d, x[0], 0xd76aa478u, 7);   SMG_MD5_STEP(SMG_F, d, a, b, c, x[1], 0xe8c7b756u, 12);
        SMG_MD5_STEP(SMG_F, c, d, a, b, x[2], 0x242070dbu, 17);  SMG_MD5_STEP(SMG_F, b, c, d, a, x[3], 0xc1bdceeeu, 22);
        SMG_MD5_STEP(SMG_F, a, b, c, d, x[4], 0xf57c0fafu, 7);   SMG_MD5_STEP(SMG_F, d, a, b, c, x[5], 0x4787c62au, 12);
        SMG_MD5_STEP(SMG_F, c, d, a, b, x[6], 0xa8304613u, 17);  SMG_MD5_STEP(SMG_F, b, c, d, a, x[7], 0xfd469501u, 22);
        SMG_MD5_STEP(SMG_F, a, b, c, d, x[8], 0x698098d8u, 7);   SMG_MD5_STEP(SMG_F, d, a, b, c, x[9], 0x8b44f7afu, 12);
        SMG_MD5_STEP(SMG_F, c, d, a, b, x[10], 0xffff5bb1u, 17); SMG_MD5_STEP(SMG_F, b, c, d, a, x[11], 0x895cd7beu, 22);
        SMG_MD5_STEP(SMG_F, a, b, c, d, x[12], 0x6b901122u, 7);  SMG_MD5_STEP(SMG_F, d, a, b, c, x[13], 0xfd987193u, 12);
        SMG_MD5_STEP(SMG_F, c, d, a, b, x[14], 0xa679438eu, 17); SMG_MD5_STEP(SMG_F, b, c, d, a, x[15], 0x49b40821u, 22);
        SMG_MD5_STEP(SMG_G, a, b, c, d, x[1], 0xf61e2562u, 5);   SMG_MD5_STEP(SMG_G, d, a, b, c, x[6], 0xc040b340u, 9);
        SMG_MD5_STEP(SMG_G, c, d, a, b, x[11], 0x265e5a51u, 14); SMG_MD5_STEP(SMG_G, b, c, d, a, x[0], 0xe9b6c7aau, 20);
        SMG_MD5_STEP(SMG_G, a, b, c, d, x[5], 0xd62f105du, 5);   SMG_MD5_STEP(SMG_G, d, a, b, c, x[10], 0x02441453u, 9);
        SMG_MD5_STEP(SMG_G, c, d, a, b, x[15], 0xd8a1e681u, 14); SMG_MD5_STEP(SMG_G, b, c, d, a, x[4], 0xe7d3fbc8u, 20);
        SMG_MD5_STEP(SMG_G, a, b, c, d, x[9], 0x21e1cde6u, 5);   SMG_MD5_STEP(SMG_G, d, a, b, c, x[14], 0xc33707d6u, 9);
        SMG_MD5_STEP(SMG_G, c, d, a, b, x[3], 0xf4d50d87u, 14);  SMG_MD5_STEP(SMG_G, b, c, d, a, x[8], 0x455a14edu, 20);
        SMG_MD5_STEP(SMG_G, a, b, c, d, x[13], 0xa9e3e905u, 5);  SMG_MD5_STEP(SMG_G, d, a, b, c, x[2], 0xfcefa3f8u, 9);
        SMG_MD5_STEP(SMG_G, c, d, a, b, x[7], 0x676f02d9u, 14);  SMG_MD5_STEP(SMG_G, b, c, d, a, x[12], 0x8d2a4c8au, 20);
        SMG_MD5_STEP(SMG_H, a, b, c, d, x[5], 0xfffa3942u, 4);   SMG_MD5_STEP(SMG_H, d, a, b, c, x[8], 0x8771f681u, 11);
        SMG_MD5_STEP(SMG_H, c, d, a, b, x[11], 0x6d9d6122u, 16); SMG_MD5_STEP(SMG_H, b, c, d, a, x[14], 0xfde5380cu, 23);
        SMG_MD5_STEP(SMG_H, a, b, c, d, x[1], 0xa4beea44u, 4);   SMG_MD5_STEP(SMG_H, d, a, b, c, x[4], 0x4bdecfa9u, 11);
        SMG_MD5_STEP(SMG_H, c, d, a, b, x[7], 0xf6bb4b60u, 16);  SMG_MD5_STEP(SMG_H, b, c, d, a, x[10], 0xbebfbc70u, 23);
        SMG_MD5_STEP(SMG_H, a, b, c, d, x[13], 0x289b7ec6u, 4);  SMG_MD5_STEP(SMG_H, d, a, b, c, x[0], 0xeaa127fau, 11);
        SMG_MD5_STEP(SMG_H, c, d, a, b, x[3], 0xd4ef3085u, 16);  SMG_MD5_STEP(SMG_H, b, c, d, a, x[6], 0x04881d05u, 23);
        SMG_MD5_STEP(SMG_H, a, b, c, d, x[9], 0xd9d4d039u, 4);   SMG_MD5_STEP(SMG_H, d, a, b, c, x[12], 0xe6db99e5u, 11);
        SMG_MD5_STEP(SMG_H, c, d, a, b, x[15], 0x1fa27cf8u, 16); SMG_MD5_STEP(SMG_H, b, c, d, a, x[2], 0xc4ac5665u, 23);
        SMG_MD5_STEP(SMG_I, a, b, c, d, x[0], 0xf4292244u, 6);   SMG_MD5_STEP(SMG_I, d, a, b, c, x[7], 0x432aff97u, 10);
        SMG_MD5_STEP(SMG_I, c, d, a, b, x[14], 0xab9423a7u, 15); SMG_MD5_STEP(SMG_I, b, c, d, a, x[5], 0xfc93a039u, 21);
        SMG_MD5_STEP(SMG_I, a, b, c, d, x[12], 0x655b59c3u, 6);  SMG_MD5_STEP(SMG_I, d, a, b, c, x[3], 0x8f0ccc92u, 10);
        SMG_MD5_STEP(SMG_I, c, d, a, b, x[10], 0xffeff47du, 15); SMG_MD5_STEP(SMG_I, b, c, d, a, x[1], 0x85845dd1u, 21);
        SMG_MD5_STEP(SMG_I, a, b, c, d, x[8], 0x6fa87e4fu, 6);   SMG_MD5_STEP(SMG_I, d, a, b, c, x[15], 0xfe2ce6e0u, 10);
        SMG_MD5_STEP(SMG_I, c, d, a, b, x[6], 0xa3014314u, 15);  SMG_MD5_STEP(SMG_I, b, c, d, a, x[13], 0x4e0811a1u, 21);
        SMG_MD5_STEP(SMG_I, a, b, c, d, x[4], 0xf7537e82u, 6);   SMG_MD5_STEP(SMG_I, d, a, b, c, x[11], 0xbd3af235u, 10);
        SMG_MD5_STEP(SMG_I, c, d, a, b, x[2], 0x2ad7d2bbu, 15);  SMG_MD5_STEP(SMG_I, b, c, d, a, x[9], 0xeb86d391u, 21);
#undef SMG_MD5_STEP
#undef SMG_F
#undef SMG_G
#undef SMG_H
#undef SMG_I
        a_ += a; b_ += b; c_ += c; d_ += d;
    }

    uint32_t a_, b_, c_, d_;
    uint64_t total_;
    size_t fill_;
    uint8_t buf_[64];
};

}  // namespace smg
