/*
 * cryo_oracle.h -- CPU ORACLE for the cryo-block codec path.
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path
 * (pg_cryogen_amd/, include/) never links, loads or calls anything here.
 *
 * What it restates: the arithmetic behind the reference's codec boundary
 *   compression.c:61-77   lz4_compress   -> LZ4_compress_fast(src,dst,B,LZ4_compressBound(B),accel)
 *   compression.c:79-91   lz4_decompress -> LZ4_decompress_safe(src,dst,csize,B)
 *   compression.c:93-109  zstd_compress  -> ZSTD_compress(dst,ZSTD_compressBound(B),src,B,level)
 *   compression.c:111-123 zstd_decompress-> ZSTD_decompress(dst,B,src,csize)
 * The reference itself holds no codec arithmetic: it links the un-vendored,
 * un-pinned third-party libraries liblz4 / libzstd (reference Makefile:5).
 * Pinned versions for encoder byte-parity: liblz4 1.9.3, libzstd 1.4.8/1.4.9
 * (the versions installed in the build image).  The algorithms restated are
 * the published LZ4 block format and the Zstandard format (RFC 8878), plus the
 * greedy LZ4 "fast" parser and the zstd "fast" strategy as those library
 * versions implement them.
 *
 * Pinning: tests/golden/ holds vectors produced by the real liblz4 1.9.3 /
 * libzstd 1.4.9 on this generator's blocks (tests/golden/make_golden.py is the
 * generating script); tests/test_oracle_*.py check every function here against
 * them and, where the libraries can be dlopen'ed, against the live libraries.
 */
#ifndef CRYO_ORACLE_H
#define CRYO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* synthetic block generator (spec: include/cryo_synth.h) */
void cryo_oracle_synth_block(uint64_t seed, uint64_t block_index, uint32_t block_size, int dist,
                             uint8_t *out);

/* LZ4 block codec (liblz4 1.9.3 semantics) */
size_t cryo_oracle_lz4_bound(size_t n);
/* returns compressed size, 0 on failure (dst too small for the bound path) */
size_t cryo_oracle_lz4_compress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, int accel);
/* returns decoded size (>=0) or -1 on malformed input, like LZ4_decompress_safe */
long cryo_oracle_lz4_decompress(const uint8_t *src, size_t csize, uint8_t *dst, size_t cap);

/* zstd (libzstd 1.4.x semantics) */
size_t cryo_oracle_zstd_bound(size_t n);
/* returns decoded size or -1 on malformed input, like ZSTD_decompress */
long cryo_oracle_zstd_decompress(const uint8_t *src, size_t csize, uint8_t *dst, size_t cap);
/* returns frame size; every level -5…22 (all nine strategies, see zstd_enc_oracle.c; sources up to 1 MiB at the
 * optimal-parser levels), 0 if unsupported */
size_t cryo_oracle_zstd_compress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, int level);

#ifdef __cplusplus
}
#endif
#endif
