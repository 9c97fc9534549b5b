/*
 * cryo_pg_rel.c -- CryoRelOps (pg_cryogen_amd/host/staging.h) over PostgreSQL's buffer manager, Generic WAL
 * and visibility map: what lets the batch write/read staging (cryo_stage_write_batch, cryo_read_data_batch,
 * cryo_scan_next_batch) run inside a backend.  SURVEY.md row f-4.
 *
 * Compiled only in a PGXS build (-DCRYO_HAVE_POSTGRES, pg/Makefile); the development image has no PostgreSQL
 * headers, so this file has only been syntax-checked there (tests/test_pg_syntax.py, declaration-only stand-ins),
 * never built.  It restates, call for call, what the reference does
 * around its one-block-at-a-time codec calls:
 *
 *   page_for_write   reference pg_cryogen.c:757-770   ReadBuffer + LockBuffer(EXCLUSIVE) + GenericXLogStart +
 *                                                    GenericXLogRegisterBuffer(GENERIC_XLOG_FULL_IMAGE)
 *   page_done        reference pg_cryogen.c:798-805   PageSetChecksumInplace + MarkBufferDirty + GenericXLogFinish,
 *                    and :823-824                     UnlockReleaseBuffer
 *   extend           reference pg_cryogen.c:745-756   P_NEW under LockRelationForExtension
 *   read_page        reference cache.c:112-113,168-169 ReadBuffer + BufferGetPage (pin only: the reference reads
 *                                                    cryo pages without a content lock)
 *   release_page     reference cache.c:118,129,161    ReleaseBuffer
 *   all_frozen       reference cache.c:145-149        visibilitymap_get_status & VISIBILITYMAP_ALL_FROZEN
 *   nblocks          reference pg_cryogen.c (RelationGetNumberOfBlocks, e.g. :1226)
 *
 * One difference from the reference, on purpose: a page is WAL-logged and released as soon as it is complete
 * (page_done) instead of all buffers of a chain being held until the end of cryo_preserve; chains of a write-behind
 * batch can be tens of thousands of pages, more than a backend may keep pinned.  So that a concurrent scan never finds
 * a valid first page in front of continuation pages that are still empty, staging.c completes the pages of a chain
 * last page first (the first page is what makes the chain reachable).
 */
#ifdef CRYO_HAVE_POSTGRES
#include "postgres.h"

#include "access/generic_xlog.h"
#include "access/visibilitymap.h"
#include "storage/bufmgr.h"
#include "storage/bufpage.h"
#include "storage/lmgr.h"
#include "utils/rel.h"

#include "cryo_pg_rel.h"

static BlockNumber
pg_nblocks(void *h)
{
    return RelationGetNumberOfBlocks(((CryoPgRel *) h)->rel);
}

static const char *
pg_read_page(void *h, BlockNumber b)
{
    CryoPgRel  *r = h;
    int         i;

    for (i = 0; i < CRYO_PG_PINS; i++)
        if (!BufferIsValid(r->pin_buf[i]))
        {
            r->pin_buf[i] = ReadBuffer(r->rel, b);
            r->pin_block[i] = b;
            return (const char *) BufferGetPage(r->pin_buf[i]);
        }
    elog(ERROR, "pg_cryogen: too many pages pinned by one chain walk");
    return NULL;
}

static void
pg_release_page(void *h, BlockNumber b)
{
    CryoPgRel  *r = h;
    int         i;

    for (i = 0; i < CRYO_PG_PINS; i++)
        if (BufferIsValid(r->pin_buf[i]) && r->pin_block[i] == b)
        {
            ReleaseBuffer(r->pin_buf[i]);
            r->pin_buf[i] = InvalidBuffer;
            return;
        }
}

static char *
pg_page_for_write(void *h, BlockNumber b)
{
    CryoPgRel  *r = h;

    Assert(!BufferIsValid(r->wbuf));
    r->wbuf = ReadBuffer(r->rel, b);
    LockBuffer(r->wbuf, BUFFER_LOCK_EXCLUSIVE);
    r->xlog = GenericXLogStart(r->rel);
    return (char *) GenericXLogRegisterBuffer(r->xlog, r->wbuf, GENERIC_XLOG_FULL_IMAGE);
}

static void
pg_page_done(void *h, BlockNumber b, char *page)
{
    CryoPgRel  *r = h;

    PageSetChecksumInplace((Page) page, b);     /* reference pg_cryogen.c:798 */
    MarkBufferDirty(r->wbuf);
    GenericXLogFinish(r->xlog);
    UnlockReleaseBuffer(r->wbuf);
    r->wbuf = InvalidBuffer;
    r->xlog = NULL;
}

static BlockNumber
pg_extend(void *h)
{
    CryoPgRel  *r = h;
    Buffer      buf;
    BlockNumber b;

    LockRelationForExtension(r->rel, ExclusiveLock);
    buf = ReadBuffer(r->rel, P_NEW);
    b = BufferGetBlockNumber(buf);
    ReleaseBuffer(buf);
    UnlockRelationForExtension(r->rel, ExclusiveLock);
    return b;
}

static bool
pg_all_frozen(void *h, BlockNumber b)
{
    CryoPgRel  *r = h;
    Buffer      vmbuf = InvalidBuffer;
    uint8       flags = visibilitymap_get_status(r->rel, b, &vmbuf);

    if (BufferIsValid(vmbuf))
        ReleaseBuffer(vmbuf);
    return (flags & VISIBILITYMAP_ALL_FROZEN) != 0;
}

static const CryoRelOps cryo_pg_ops = {
    pg_nblocks, pg_read_page, pg_page_for_write, pg_extend, pg_all_frozen, pg_page_done, pg_release_page
};

/* bind an open relation; `store` is caller memory that lives as long as the CryoRel is used (e.g. in the scan or
 * modify state: reference CryoScanDesc / CryoModifyState) */
void
cryo_pg_bind(Relation rel, CryoPgRel *store, CryoRel *out)
{
    int         i;

    memset(store, 0, sizeof *store);
    store->rel = rel;
    store->wbuf = InvalidBuffer;
    for (i = 0; i < CRYO_PG_PINS; i++)
        store->pin_buf[i] = InvalidBuffer;
    out->relid = RelationGetRelid(rel);
    out->handle = store;
    out->ops = &cryo_pg_ops;
}

/*
 * The reference's Relation-typed cache entry points (reference cache.h:25-27), so that its seven call sites in
 * pg_cryogen.c (:117,265,389,423,873,1017; cryo_init_cache at :172) compile UNCHANGED against this repo's cache.c:
 * bind the Relation for the duration of the call.  Every page a chain walk pins is released before the walk returns
 * (release_page); after an elog(ERROR) the resource owner drops what is left, as for the reference's own ReadBuffer.
 */
CryoError
cryo_read_data(Relation rel, SeqScanIterator *iter, BlockNumber block, CacheEntry *result)
{
    CryoPgRel   store;
    CryoRel     r;

    cryo_pg_bind(rel, &store, &r);
    return cryo_read_data_rel(&r, iter, block, result);
}

CacheEntry
cryo_cache_allocate(Relation rel, BlockNumber blockno)
{
    CryoPgRel   store;
    CryoRel     r;

    cryo_pg_bind(rel, &store, &r);
    return cryo_cache_allocate_rel(&r, blockno);
}

/*
 * Beyond the drop-in.  Reads need nothing: cryo_read_data() above reaches cryo_read_data_rel(), whose miss path reads
 * ahead in a copy of the scan's iterator (pg_cryogen.gpu_readahead_blocks), so the unchanged cryo_getnextslot
 * (reference pg_cryogen.c:262-277) decodes K blocks per codec call.  Writes: one call-site change a maintainer may make
 * (nothing else of pg_cryogen.c changes):
 *
 *   write-behind, replacing the per-block cryo_preserve() of cryo_multi_insert (reference pg_cryogen.c:603-663):
 *       keep K full cryo blocks (state->data) in a backend-local array instead of compressing each when it
 *       fills; at K blocks or at finish_bulk_insert:
 *           CryoPgRel s; CryoRel r; cryo_pg_bind(rel, &s, &r);
 *           cryo_stage_write_batch(&r, blocks, K, compression_method_guc, GetCurrentTransactionId(), first_blocks);
 *       then the metapage update of reference pg_cryogen.c:807-821 once for the batch.
 */
#endif /* CRYO_HAVE_POSTGRES */
