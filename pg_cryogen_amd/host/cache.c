/* cache.c -- see cache.h */
#include "cache.h"
#include "scan_iterator.h"

typedef struct
{
    Oid relid;
    BlockNumber blockno;
    bool pinned;     /* insert buffer: cannot be evicted */
    uint64 ts;       /* 0 = free; otherwise last-use tick (LRU) */
    uint32 nblocks;  /* PostgreSQL pages of the chain */
    TransactionId xid;
    BlockNumber *blocks;
    char *data;      /* cryo_blcksz bytes */
} Slot;

static Slot *slots;
static int nslots;
static uint32 max_chain;
static uint64 tick, n_hits, n_misses, n_codec_calls;

void cryo_cache_shutdown(void)
{
    int i;
    for (i = 0; i < nslots; i++) { free(slots[i].data); free(slots[i].blocks); }
    free(slots);
    slots = NULL;
    nslots = 0;
}

int cryo_cache_configure(int n)
{
    /* sized at _PG_init (cryo_init_cache): nothing here may open the GPU */
    const Size ba = cryo_host_codec_bound(COMP_LZ4, cryo_blcksz), bb = cryo_host_codec_bound(COMP_ZSTD, cryo_blcksz);
    const Size worst = ba > bb ? ba : bb;
    int i;
    cryo_cache_shutdown();
    if (n < 1) return -1;
    slots = calloc((size_t)n, sizeof *slots);
    if (!slots) return -1;
    max_chain = (uint32)cryo_pages_needed(worst);
    for (i = 0; i < n; i++) {
        slots[i].data = malloc(cryo_blcksz);
        slots[i].blocks = malloc(max_chain * sizeof(BlockNumber));
        if (!slots[i].data || !slots[i].blocks) { nslots = i + 1; cryo_cache_shutdown(); return -1; }
    }
    nslots = n;
    tick = n_hits = n_misses = n_codec_calls = 0;
    return 0;
}

void cryo_init_cache(void) { (void)cryo_cache_configure(16); }

static int find_slot(Oid relid, BlockNumber blockno)
{
    int i;
    for (i = 0; i < nslots; i++)
        if (slots[i].ts != 0 && slots[i].relid == relid && slots[i].blockno == blockno) return i;
    return InvalidCacheEntry;
}

/* a free slot, else the least recently used unpinned one; InvalidCacheEntry if all pinned */
static int allocate_slot(void)
{
    int i, victim = InvalidCacheEntry;
    uint64 min_ts = ~(uint64)0;
    for (i = 0; i < nslots; i++) {
        if (slots[i].pinned) continue;
        if (slots[i].ts == 0) return i;
        if (slots[i].ts < min_ts) { min_ts = slots[i].ts; victim = i; }
    }
    if (victim != InvalidCacheEntry)
        elog(DEBUG1, "pg_cryogen: evicted cache entry for (%u, %u)", slots[victim].relid, slots[victim].blockno);
    return victim;
}

/* unpinned slots: what one batch can be given without evicting its own members */
static int evictable_slots(void)
{
    int i, n = 0;
    for (i = 0; i < nslots; i++) if (!slots[i].pinned) n++;
    return n;
}

/* iter: the scan's iterator -- pages of chains read or served from the cache are struck from it (reference cache.c:174,
 * 235-242); ahead: entries from this index on strike theirs from `iter_ahead` instead (the read-ahead of cryo_read_data_rel
 * looks ahead in a copy of the iterator: what the scan has not asked for yet stays in the scan's own set) */
static CryoError load_blocks(CryoRel *rel, SeqScanIterator *iter, const BlockNumber *blocks, int k,
                             CacheEntry *results, CryoError *errors, int ahead, SeqScanIterator *iter_ahead)
{
    /* gather the chains of every missing block, then ONE decompress call per method.
     *
     * Every slot a batch hands out -- hit or miss -- is pinned until the batch returns, so a later miss of
     * the same batch cannot evict it (a batch larger than the cache gets CRYO_ERR_CACHE_IS_FULL for the
     * members that do not fit, never another block's data).  A chain is read into temporaries and a victim
     * is only chosen, and overwritten, once the read succeeded: a failed probe (continuation page of a
     * bitmap scan, broken chain) leaves the cache as it was (reference cache.c:184-233 removes the victim
     * from its hash before reuse; same effect). */
    const CryoCodecOps *ops = cryo_host_codec_ops();
    char **comp = calloc((size_t)k, sizeof *comp);
    uint32_t *csz = calloc((size_t)k, sizeof *csz);
    int *slot_of = malloc((size_t)k * sizeof *slot_of);   /* slot this entry loads (miss), else Invalid */
    int *pinned_by = malloc((size_t)k * sizeof *pinned_by); /* slot this entry pinned for the batch, else Invalid */
    CompressionMethod *meth = calloc((size_t)k, sizeof *meth);
    BlockNumber *tmp_blocks = malloc((size_t)(max_chain ? max_chain : 1) * sizeof *tmp_blocks);
    CryoError first_err = CRYO_ERR_SUCCESS;
    int i, m;
    if (!comp || !csz || !slot_of || !pinned_by || !meth || !tmp_blocks) {
        free(comp); free(csz); free(slot_of); free(pinned_by); free(meth); free(tmp_blocks);
        return CRYO_ERR_CACHE_IS_FULL;
    }

    for (i = 0; i < k; i++) {
        int s;
        SeqScanIterator *it = i >= ahead ? iter_ahead : iter;
        slot_of[i] = pinned_by[i] = InvalidCacheEntry;
        errors[i] = CRYO_ERR_SUCCESS;
        if (rel->ops->nblocks(rel->handle) <= blocks[i] || blocks[i] == CRYO_META_PAGE) {
            errors[i] = CRYO_ERR_WRONG_STARTING_BLOCK; results[i] = InvalidCacheEntry; continue;
        }
        s = find_slot(rel->relid, blocks[i]);
        if (s != InvalidCacheEntry) {
            /* cached, or claimed by an earlier entry of this batch (a repeated block number) */
            uint32 j;
            n_hits++; slots[s].ts = ++tick; results[i] = s;
            if (!slots[s].pinned) { slots[s].pinned = true; pinned_by[i] = s; }
            /* the cached block's pages must not be handed out again (cache.c:235-242,290-292) */
            for (j = 0; j < slots[s].nblocks; j++) cryo_seqscan_iter_exclude(it, slots[s].blocks[j], true);
            continue;
        }
        n_misses++;
        {
            Size cs = 0;
            uint32 nb = 0, j;
            TransactionId xid = 0;
            Slot *sl;
            errors[i] = cryo_stage_read_chain(rel, blocks[i], &comp[i], &cs, &meth[i], &xid, tmp_blocks, max_chain, &nb);
            if (errors[i] != CRYO_ERR_SUCCESS) { results[i] = InvalidCacheEntry; continue; }
            s = allocate_slot();
            if (s == InvalidCacheEntry) { errors[i] = CRYO_ERR_CACHE_IS_FULL; results[i] = InvalidCacheEntry; continue; }
            sl = &slots[s];
            /* continuation pages are not block starts (cache.c:174).  miss_ok: the reference
             * passes false here and its regression output pins the resulting internal error
             * (expected/pg_cryogen.out:166); a page the iterator already handed out is
             * harmless (it reads as WRONG_STARTING_BLOCK), so it is tolerated here */
            for (j = 1; j < nb; j++) cryo_seqscan_iter_exclude(it, tmp_blocks[j], true);
            csz[i] = (uint32_t)cs;
            /* claim the slot now (pinned for the duration of the batch so a later miss cannot evict it) */
            memcpy(sl->blocks, tmp_blocks, (size_t)nb * sizeof *tmp_blocks);
            sl->nblocks = nb; sl->xid = xid;
            sl->relid = rel->relid; sl->blockno = blocks[i]; sl->ts = ++tick; sl->pinned = true;
            slot_of[i] = pinned_by[i] = s;
            results[i] = s;
        }
    }
    for (m = COMP_LZ4; m <= COMP_ZSTD; m++) {
        int cnt = 0, j = 0;
        for (i = 0; i < k; i++) if (slot_of[i] != InvalidCacheEntry && (int)meth[i] == m) cnt++;
        if (!cnt) continue;
        {
            const void **srcs = malloc((size_t)cnt * sizeof *srcs);
            uint32_t *sz = malloc((size_t)cnt * sizeof *sz);
            int32_t *st = calloc((size_t)cnt, sizeof *st);
            int *idx = malloc((size_t)cnt * sizeof *idx);
            char **outs = malloc((size_t)cnt * sizeof *outs);
            uint64_t *keys = malloc((size_t)cnt * sizeof *keys);
            int rc = -1;
            if (srcs && sz && st && idx && outs && keys && ops) {
                for (i = 0; i < k; i++)
                    if (slot_of[i] != InvalidCacheEntry && (int)meth[i] == m) {
                        srcs[j] = comp[i]; sz[j] = csz[i]; idx[j] = i; outs[j] = slots[slot_of[i]].data;
                        keys[j] = ((uint64_t)rel->relid << 32) | blocks[i]; /* the cache's own key, reference cache.c:37-47 */
                        j++;
                    }
                /* decoded blocks land in their cache slots directly (no second copy); a codec that keeps decoded
                 * blocks in device memory is told who they are */
                if (ops->decompress_blocks_keyed)
                    rc = ops->decompress_blocks_keyed(ops->ctx, m, keys, srcs, sz, (size_t)cnt, (void *const *)outs, cryo_blcksz, st);
                else if (ops->decompress_blocks_scatter)
                    rc = ops->decompress_blocks_scatter(ops->ctx, m, srcs, sz, (size_t)cnt, (void *const *)outs, cryo_blcksz, st);
                else {
                    char *out = malloc((size_t)cnt * cryo_blcksz);
                    if (out) {
                        rc = ops->decompress_blocks(ops->ctx, m, srcs, sz, (size_t)cnt, out, cryo_blcksz, st);
                        for (j = 0; j < cnt; j++)
                            if (rc == 0 && st[j] == 0) memcpy(outs[j], out + (size_t)j * cryo_blcksz, cryo_blcksz);
                        free(out);
                    }
                }
                n_codec_calls++;
            }
            for (i = 0, j = 0; i < k; i++) {
                if (slot_of[i] == InvalidCacheEntry || (int)meth[i] != m) continue;
                if (!(rc == 0 && st && st[j] == 0)) {
                    errors[i] = CRYO_ERR_DECOMPRESSION_FAILED; results[i] = InvalidCacheEntry; slots[slot_of[i]].ts = 0;
                }
                j++;
            }
            free(srcs); free(sz); free(st); free(idx); free(outs); free(keys);
        }
    }
    for (i = 0; i < k; i++) {
        /* a repeated block number shares the fate of its first occurrence */
        if (errors[i] == CRYO_ERR_SUCCESS && results[i] != InvalidCacheEntry && slots[results[i]].ts == 0) {
            errors[i] = CRYO_ERR_DECOMPRESSION_FAILED; results[i] = InvalidCacheEntry;
        }
    }
    for (i = 0; i < k; i++) {
        if (pinned_by[i] != InvalidCacheEntry) slots[pinned_by[i]].pinned = false;
        free(comp[i]);
        if (errors[i] != CRYO_ERR_SUCCESS && first_err == CRYO_ERR_SUCCESS) first_err = errors[i];
    }
    free(comp); free(csz); free(slot_of); free(pinned_by); free(meth); free(tmp_blocks);
    return first_err;
}

CryoError cryo_read_data_batch(CryoRel *rel, const BlockNumber *blocks, int k, CacheEntry *results, CryoError *errors)
{
    if (k <= 0) return CRYO_ERR_SUCCESS;
    return load_blocks(rel, NULL, blocks, k, results, errors, k, NULL);
}

/* strikes the continuation pages of the chain that starts at page b from `iter` (header looks only, no decode) */
static void exclude_chain(CryoRel *rel, SeqScanIterator *iter, BlockNumber b, BlockNumber next)
{
    uint32 guard = 0;
    while (BlockNumberIsValid(next) && next < rel->ops->nblocks(rel->handle) && guard++ < max_chain) {
        const CryoPageHeader *q;
        BlockNumber nn;
        bool mine;
        cryo_seqscan_iter_exclude(iter, next, true);
        q = (const CryoPageHeader *)rel->ops->read_page(rel->handle, next);
        if (!q) break;
        mine = q->first == b;
        nn = q->next;
        if (rel->ops->release_page) rel->ops->release_page(rel->handle, next);
        if (!mine) break;
        next = nn;
    }
}

/* pops page numbers from `iter` until `want` of them start a chain (cheap header looks); pages that are empty or belong to
 * another chain are skipped, every chain's continuation pages are struck from `iter`.  *eof: the relation ended first. */
static int next_block_starts(CryoRel *rel, SeqScanIterator *iter, int want, BlockNumber *cand, bool *eof)
{
    int nc = 0;
    *eof = false;
    while (nc < want) {
        const BlockNumber b = cryo_seqscan_iter_next(iter);
        const CryoPageHeader *pg;
        BlockNumber first, next;
        bool empty;
        if (!BlockNumberIsValid(b) || b >= rel->ops->nblocks(rel->handle)) { *eof = true; break; }
        pg = (const CryoPageHeader *)rel->ops->read_page(rel->handle, b);
        if (!pg) continue;
        empty = pg->base.pd_upper == 0; first = pg->first; next = pg->next;
        if (rel->ops->release_page) rel->ops->release_page(rel->handle, b);
        if (empty || first != b) continue; /* empty page / continuation page */
        exclude_chain(rel, iter, b, next); /* now, so that they are not offered as candidates */
        cand[nc++] = b;
    }
    return nc;
}

/*
 * The reference's entry point (cache.c:244-297): one block per call, which is all the unchanged table AM ever asks for
 * (pg_cryogen.c:262-265).  A device call pays off from a handful of blocks on (crossover table), so a
 * MISS of a sequential scan -- the AM passes its iterator -- also loads the next block starts the scan is going to ask
 * for, up to pg_cryogen.gpu_readahead_blocks in all (and never more than half the slots that may be evicted), with the ONE
 * codec call load_blocks makes per method; the scan's next calls are hits.  The look-ahead walks a COPY of the iterator:
 * the scan's own set changes exactly as it would without read-ahead (the block asked for loses its continuation pages,
 * nothing else), and a scan that stops early (LIMIT) has wasted at most K - 1 decodes.
 */
CryoError cryo_read_data_rel(CryoRel *rel, void *iter_, BlockNumber block, CacheEntry *result)
{
    SeqScanIterator *iter = iter_;
    CryoError err = CRYO_ERR_SUCCESS;
    int k = cryo_gpu_readahead_blocks_guc;
    *result = InvalidCacheEntry;
    {
        const int room = evictable_slots() / 2;
        if (k > room) k = room;
    }
    if (iter && k > 1 && block != CRYO_META_PAGE && block < rel->ops->nblocks(rel->handle) &&
        find_slot(rel->relid, block) == InvalidCacheEntry) {
        const CryoPageHeader *pg = (const CryoPageHeader *)rel->ops->read_page(rel->handle, block);
        BlockNumber first = InvalidBlockNumber, next = InvalidBlockNumber;
        bool start = false;
        if (pg) {
            start = pg->base.pd_upper != 0 && pg->first == block;
            first = pg->first; next = pg->next;
            if (rel->ops->release_page) rel->ops->release_page(rel->handle, block);
        }
        (void)first;
        if (start) {
            SeqScanIterator *look = cryo_seqscan_iter_clone(iter);
            BlockNumber *cand = malloc((size_t)k * sizeof *cand);
            CacheEntry *res = malloc((size_t)k * sizeof *res);
            CryoError *errs = malloc((size_t)k * sizeof *errs);
            if (look && cand && res && errs) {
                bool eof;
                int nc = 1, i;
                cand[0] = block;
                exclude_chain(rel, look, block, next);
                nc += next_block_starts(rel, look, k - 1, cand + 1, &eof);
                (void)load_blocks(rel, iter, cand, nc, res, errs, 1, look);
                /* what went wrong with a block nobody asked for yet is reported when it is asked for */
                for (i = 1; i < nc; i++) (void)errs[i];
                *result = res[0];
                err = errs[0];
                cryo_seqscan_iter_free(look); free(cand); free(res); free(errs);
                return err;
            }
            cryo_seqscan_iter_free(look); free(cand); free(res); free(errs);
        }
    }
    (void)load_blocks(rel, iter, &block, 1, result, &err, 1, NULL);
    return err;
}

/*
 * Read-ahead for a sequential scan: pop block starts from the iterator in the reference's
 * order (lowest unread first), skip pages that are not block starts or are empty, gather up
 * to k chains and decode them with one codec call per method.  Returns the number of blocks
 * delivered in starts[]/entries[] (0 = end of relation).
 */
int cryo_scan_next_batch(CryoRel *rel, void *iter_, int k, BlockNumber *starts, CacheEntry *entries,
                         CryoError *errors)
{
    SeqScanIterator *iter = iter_;
    BlockNumber *cand = malloc((size_t)k * sizeof *cand);
    CacheEntry *res = malloc((size_t)k * sizeof *res);
    CryoError *errs = malloc((size_t)k * sizeof *errs);
    int got = 0;
    if (!cand || !res || !errs || k <= 0) { free(cand); free(res); free(errs); return 0; }
    /* a batch pins every slot it hands out: never pop more block starts than there are evictable slots, or
     * the surplus would be lost to the scan as CRYO_ERR_CACHE_IS_FULL (the caller comes back for the rest) */
    {
        const int room = evictable_slots();
        if (k > room) k = room > 0 ? room : 1;
    }
    while (got < k) {
        int want = k - got, nc, i;
        bool eof;
        nc = next_block_starts(rel, iter, want, cand, &eof);
        if (eof) want = nc;
        if (nc == 0) break;
        (void)load_blocks(rel, iter, cand, nc, res, errs, nc, NULL);
        for (i = 0; i < nc; i++) { starts[got] = cand[i]; entries[got] = res[i]; errors[got] = errs[i]; got++; }
        if (want < k - (got - nc)) break; /* hit the end of the relation */
    }
    free(cand); free(res); free(errs);
    return got;
}

#ifndef CRYO_HAVE_POSTGRES
/* PostgreSQL-free builds: the reference's names on the CryoRel (a PGXS build gets Relation-typed ones, pg/cryo_pg_rel.c) */
CryoError cryo_read_data(CryoRel *rel, void *iter, BlockNumber block, CacheEntry *result)
{
    return cryo_read_data_rel(rel, iter, block, result);
}
CacheEntry cryo_cache_allocate(CryoRel *rel, BlockNumber blockno) { return cryo_cache_allocate_rel(rel, blockno); }
#endif

CacheEntry cryo_cache_allocate_rel(CryoRel *rel, BlockNumber blockno)
{
    int s = find_slot(rel->relid, blockno);
    if (s == InvalidCacheEntry) {
        s = allocate_slot();
        if (s == InvalidCacheEntry) {
            elog(ERROR, "pg_cryogen: %s", cryo_cache_err(CRYO_ERR_CACHE_IS_FULL));
            return InvalidCacheEntry;
        }
    }
    slots[s].relid = rel->relid;
    slots[s].blockno = blockno;
    slots[s].ts = ++tick;
    slots[s].pinned = true;
    slots[s].nblocks = 0;
    return s;
}

void cryo_cache_release(CacheEntry entry)
{
    if (entry < 0 || entry >= nslots) { elog(ERROR, "pg_cryogen: invalid cache entry"); return; }
    if (!slots[entry].pinned) { elog(ERROR, "pg_cryogen: trying to release read-only cache entry"); return; }
    slots[entry].ts = 0;
    slots[entry].pinned = false;
}

/* The relcache callback of reference pg_cryogen.c:163-167.  relid == InvalidOid is what PostgreSQL passes after a
 * sinval-queue reset ("anything may have changed"): every unpinned slot, and every entry of the device pool.  Uses the
 * codec binding only if this backend has one already (never opens the GPU from inside an invalidation). */
void cryo_cache_invalidate_relation(Oid relid)
{
    const CryoCodecOps *ops = cryo_host_codec_ops_if_open();
    int i;
    for (i = 0; i < nslots; i++)
        if ((relid == InvalidOid || slots[i].relid == relid) && !slots[i].pinned) slots[i].ts = 0;
    if (ops && ops->pool_invalidate) ops->pool_invalidate(ops->ctx, (uint32_t)relid); /* the device-resident copies too */
}

uint32 cryo_cache_get_pg_nblocks(CacheEntry entry) { return slots[entry].nblocks; }
char *cryo_cache_get_data(CacheEntry entry) { slots[entry].ts = ++tick; return slots[entry].data; }
TransactionId cryo_cache_get_xid(CacheEntry entry) { return slots[entry].xid; }

const char *cryo_cache_err(CryoError err)
{
    switch (err) {
    case CRYO_ERR_SUCCESS: return "success";
    case CRYO_ERR_WRONG_STARTING_BLOCK: return "wrong starting block number";
    case CRYO_ERR_DECOMPRESSION_FAILED: return "decompression failed";
    case CRYO_ERR_EMPTY_BLOCK: return "empty block";
    case CRYO_ERR_CACHE_IS_FULL: return "cannot allocate cache slot; all slots are locked for modification";
    default: return "unknown error";
    }
}

uint64 cryo_cache_hits(void) { return n_hits; }
uint64 cryo_cache_misses(void) { return n_misses; }
uint64 cryo_cache_codec_calls(void) { return n_codec_calls; }
