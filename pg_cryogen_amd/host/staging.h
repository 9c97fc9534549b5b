/*
 * staging.h -- block write/read staging: cryo block <-> chain of 8 KiB PostgreSQL pages.
 *
 * Rewrites, PostgreSQL-free and batch-capable, the page-chain logic of
 *   cryo_pages_needed / cryo_preserve      reference pg_cryogen.c:692-827  (write)
 *   cryo_read_decompress (chain walk part) reference cache.c:100-176       (read)
 * producing and consuming byte-identical pages: CryoPageHeader{first,next} (32 B) on every
 * page, CryoFirstPageHeader{created_xid, compression_method, compressed_size, npages} (48 B)
 * on the first, payload 8144 / 8160 bytes, pd_lower = header + payload, pd_upper = pd_special
 * = BLCKSZ (pg_cryogen.c:775-797).
 *
 * What PostgreSQL provides (buffer manager, relation extension, visibility map, WAL) is
 * reached through CryoRelOps; the in-memory implementation below is the PG-free harness
 * ("mini-AM") used by the plumbing tests, a PGXS build supplies bufmgr-backed callbacks.
 */
#ifndef CRYO_STAGING_H
#define CRYO_STAGING_H

#include "storage.h"

typedef enum
{
    CRYO_ERR_SUCCESS = 0,
    CRYO_ERR_DECOMPRESSION_FAILED,
    CRYO_ERR_WRONG_STARTING_BLOCK,
    CRYO_ERR_EMPTY_BLOCK,
    CRYO_ERR_CACHE_IS_FULL
} CryoError; /* reference cache.h:13-20 */

typedef struct CryoRelOps {
    BlockNumber (*nblocks)(void *rel);                 /* RelationGetNumberOfBlocks            */
    const char *(*read_page)(void *rel, BlockNumber b); /* ReadBuffer + BufferGetPage (read)    */
    char *(*page_for_write)(void *rel, BlockNumber b);  /* buffer locked exclusive, WAL-registered */
    BlockNumber (*extend)(void *rel);                  /* P_NEW under the extension lock       */
    bool (*all_frozen)(void *rel, BlockNumber b);      /* visibilitymap ALL_FROZEN bit         */
    /* optional (NULL in the in-memory harness): a written page is complete -- PageSetChecksumInplace, MarkBufferDirty,
     * GenericXLogFinish, UnlockReleaseBuffer (reference pg_cryogen.c:798-805,823-824) */
    void (*page_done)(void *rel, BlockNumber b, char *page);
    /* optional: a page obtained with read_page is no longer needed -- ReleaseBuffer (reference cache.c:171) */
    void (*release_page)(void *rel, BlockNumber b);
} CryoRelOps;

typedef struct CryoRel {
    Oid relid;
    void *handle;
    const CryoRelOps *ops;
} CryoRel;

/* pages a compressed block of `size` bytes occupies: 1 + ceil(max(0, size-8144)/8160)
 * (pg_cryogen.c:692-704; int instead of the reference's uint8, which wraps at 256 pages) */
int cryo_pages_needed(Size size);

/* write one compressed block as a page chain starting at the preallocated `first_block`
 * (pg_cryogen.c:747-805); the chain's block numbers are returned in blocks[0..*npages) */
int cryo_stage_write_chain(CryoRel *rel, BlockNumber first_block, CompressionMethod method, TransactionId xid,
                           const char *compressed, Size csize, BlockNumber *blocks, int max_blocks, int *npages);

/* write-behind: compress K full cryo blocks with ONE codec call and emit their chains.
 * data = K consecutive blocks of cryo_blcksz bytes; first_blocks[i] is the block number
 * reserved for block i (cryo_reserve_blockno, pg_cryogen.c:588-601), or InvalidBlockNumber
 * to extend the relation. */
int cryo_stage_write_batch(CryoRel *rel, const char *data, int k, CompressionMethod method, TransactionId xid,
                           BlockNumber *first_blocks);

/* walk one chain and reassemble the compressed bytes (cache.c:108-176).  *compressed is
 * malloc'ed (the reference leaks its palloc, cache.c:134; here the caller frees it). */
CryoError cryo_stage_read_chain(CryoRel *rel, BlockNumber block, char **compressed, Size *csize,
                                CompressionMethod *method, TransactionId *xid, BlockNumber *blocks,
                                uint32 max_blocks, uint32 *nblocks);

/* ---- in-memory relation (PG-free harness) ---- */
typedef struct CryoMemRel CryoMemRel;
CryoMemRel *cryo_memrel_create(void);
void cryo_memrel_destroy(CryoMemRel *m);
void cryo_memrel_bind(CryoMemRel *m, Oid relid, CryoRel *out);
BlockNumber cryo_memrel_reserve(CryoMemRel *m); /* like cryo_reserve_blockno: extend by one page */
void cryo_memrel_set_frozen(CryoMemRel *m, BlockNumber b, bool frozen);
const char *cryo_memrel_page(CryoMemRel *m, BlockNumber b);
BlockNumber cryo_memrel_nblocks(CryoMemRel *m);

#endif
