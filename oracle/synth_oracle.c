/*
 * synth_oracle.c -- CPU ORACLE side of the synthetic cryo-block generator
 * (test infrastructure, see cryo_oracle.h).  The byte-level spec lives in
 * include/cryo_synth.h; the block layout it produces is the one built by
 * cryo_init_page()/cryo_storage_insert() (reference storage.c:15-50).
 */
#include "cryo_oracle.h"
#include "../include/cryo_synth.h"

void cryo_oracle_synth_block(uint64_t seed, uint64_t block_index, uint32_t block_size, int dist,
                             uint8_t *out)
{
    cryo_synth_geom g = cryo_synth_geometry(block_size, dist);
    uint32_t off;
    for (off = 0; off < block_size; off++)
        out[off] = cryo_synth_byte(seed, block_index, block_size, dist, g, off);
}
