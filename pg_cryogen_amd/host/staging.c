/* staging.c -- see staging.h */
#include "staging.h"

#include <stdio.h>

#define FIRST_PAYLOAD (BLCKSZ - sizeof(CryoFirstPageHeader)) /* 8144 */
#define NEXT_PAYLOAD (BLCKSZ - sizeof(CryoPageHeader))       /* 8160 */
#define MIN(a, b) ((a) < (b) ? (a) : (b))

int cryo_pages_needed(Size size)
{
    if (size <= FIRST_PAYLOAD) return 1;
    return 1 + (int)((size - FIRST_PAYLOAD + NEXT_PAYLOAD - 1) / NEXT_PAYLOAD);
}

int cryo_stage_write_chain(CryoRel *rel, BlockNumber first_block, CompressionMethod method, TransactionId xid,
                           const char *compressed, Size csize, BlockNumber *blocks, int max_blocks, int *npages_out)
{
    const int npages = cryo_pages_needed(csize);
    const char *p = compressed;
    Size left = csize;
    int i;

    if (npages > max_blocks || csize == 0) return -1;
    /* the first page was preallocated; the others extend the relation (pg_cryogen.c:747-756) */
    blocks[0] = first_block;
    for (i = 1; i < npages; i++) blocks[i] = rel->ops->extend(rel->handle);

    /* The pages are completed (WAL-logged, unlocked: page_done) one by one, LAST page first: the first page is what
     * makes a chain reachable (a reader takes a page with pd_upper != 0 and first == its own number for a block
     * start), so it is written when every continuation page already is.  The reference holds all buffers of a chain
     * exclusively until the end of cryo_preserve (pg_cryogen.c:757-824); a write-behind batch cannot pin that many. */
    for (i = npages - 1; i >= 0; i--) {
        char *page = rel->ops->page_for_write(rel->handle, blocks[i]);
        CryoPageHeader *hdr = (CryoPageHeader *)page;
        const Size before = i == 0 ? 0 : FIRST_PAYLOAD + (Size)(i - 1) * NEXT_PAYLOAD; /* payload bytes on the pages before this one */
        Size hdr_size, content;
        if (!page) return -1;
        p = compressed + before;
        left = csize - before;
        memset(page, 0, BLCKSZ);
        hdr->first = blocks[0];
        hdr->next = (i + 1 < npages) ? blocks[i + 1] : InvalidBlockNumber;
        if (i == 0) {
            CryoFirstPageHeader *fh = (CryoFirstPageHeader *)page;
            fh->npages = (uint16)npages;
            fh->compression_method = method;
            fh->compressed_size = (uint32)csize;
            fh->created_xid = xid;
        }
        hdr_size = (i == 0) ? sizeof(CryoFirstPageHeader) : sizeof(CryoPageHeader);
        content = MIN((Size)BLCKSZ - hdr_size, left);
        /* pd_upper must not stay 0 or the page counts as new (pg_cryogen.c:787-794) */
        hdr->base.pd_upper = BLCKSZ;
        hdr->base.pd_lower = (uint16)(hdr_size + content);
        hdr->base.pd_special = BLCKSZ;
        memcpy(page + hdr_size, p, content);
        if (rel->ops->page_done) rel->ops->page_done(rel->handle, blocks[i], page);
    }
    *npages_out = npages;
    return 0;
}

int cryo_stage_write_batch(CryoRel *rel, const char *data, int k, CompressionMethod method, TransactionId xid,
                           BlockNumber *first_blocks)
{
    const CryoCodecOps *ops = cryo_host_codec_ops();
    Size bound, stride;
    char *comp;
    uint32_t *sizes;
    BlockNumber *chain;
    int i, rc, max_pages;

    if (!ops || k <= 0) return -1;
    bound = ops->bound((int)method, cryo_blcksz);
    stride = (bound + 15) & ~(Size)15;
    max_pages = cryo_pages_needed(bound);
    comp = malloc((Size)k * stride);
    sizes = malloc((Size)k * sizeof *sizes);
    chain = malloc((Size)max_pages * sizeof *chain);
    if (!comp || !sizes || !chain) { free(comp); free(sizes); free(chain); return -1; }
    rc = ops->compress_blocks(ops->ctx, (int)method,
                              method == COMP_LZ4 ? lz4_acceleration_guc : zstd_compression_level_guc, data,
                              cryo_blcksz, (size_t)k, comp, stride, sizes);
    for (i = 0; rc == 0 && i < k; i++) {
        int np;
        if (!BlockNumberIsValid(first_blocks[i])) first_blocks[i] = rel->ops->extend(rel->handle);
        rc = cryo_stage_write_chain(rel, first_blocks[i], method, xid, comp + (Size)i * stride, sizes[i], chain,
                                    max_pages, &np);
    }
    free(comp); free(sizes); free(chain);
    return rc;
}

CryoError cryo_stage_read_chain(CryoRel *rel, BlockNumber block, char **compressed, Size *csize_out,
                                CompressionMethod *method, TransactionId *xid, BlockNumber *blocks,
                                uint32 max_blocks, uint32 *nblocks)
{
    const BlockNumber first_block = block;
    const CryoPageHeader *page = (const CryoPageHeader *)rel->ops->read_page(rel->handle, block);
    const CryoFirstPageHeader *fh;
    Size size, csize;
    char *buf, *p;

#define RELEASE(b) do { if (rel->ops->release_page) rel->ops->release_page(rel->handle, (b)); } while (0)
    *compressed = NULL;
    *nblocks = 0;
    if (!page) return CRYO_ERR_EMPTY_BLOCK;
    if (page->base.pd_upper == 0) { RELEASE(block); return CRYO_ERR_EMPTY_BLOCK; } /* PageIsNew */
    /* a BRIN bitmap scan may ask for a block in the middle of a chain (cache.c:122-130) */
    if (page->first != block) { RELEASE(block); return CRYO_ERR_WRONG_STARTING_BLOCK; }
    fh = (const CryoFirstPageHeader *)page;
    size = csize = fh->compressed_size;
    *method = fh->compression_method;
    /* frozen blocks are flagged in the visibility map, not rewritten (cache.c:137-149) */
    *xid = rel->ops->all_frozen(rel->handle, block) ? FrozenTransactionId : fh->created_xid;
    if (csize == 0) { RELEASE(block); return CRYO_ERR_DECOMPRESSION_FAILED; }
    p = buf = malloc(csize);
    if (!buf) { RELEASE(block); return CRYO_ERR_DECOMPRESSION_FAILED; }
    if (*nblocks < max_blocks) blocks[(*nblocks)++] = block;

    for (;;) {
        const Size hdr_size = CryoPageHeaderSize(page, block);
        const Size l = MIN((Size)BLCKSZ - hdr_size, size);
        const BlockNumber cur = block;
        memcpy(p, (const char *)page + hdr_size, l);
        p += l;
        size -= l;
        block = page->next;
        RELEASE(cur);
        if (size == 0) break;
        if (!BlockNumberIsValid(block) || block >= rel->ops->nblocks(rel->handle)) break;
        page = (const CryoPageHeader *)rel->ops->read_page(rel->handle, block);
        if (!page) break;
        if (page->first != first_block) { RELEASE(block); break; } /* broken chain */
        if (*nblocks < max_blocks) blocks[(*nblocks)++] = block;
    }
#undef RELEASE
    if (size != 0) { /* chain shorter than compressed_size: the reference would decode garbage */
        free(buf);
        return CRYO_ERR_DECOMPRESSION_FAILED;
    }
    *compressed = buf;
    *csize_out = csize;
    return CRYO_ERR_SUCCESS;
}

/* ---------------- in-memory relation ---------------- */
struct CryoMemRel {
    char **pages;
    uint8 *frozen;
    BlockNumber n, cap;
};

static BlockNumber mem_nblocks(void *r) { return ((CryoMemRel *)r)->n; }
static BlockNumber mem_extend(void *r)
{
    CryoMemRel *m = r;
    if (m->n == m->cap) {
        BlockNumber nc = m->cap ? m->cap * 2 : 64;
        m->pages = realloc(m->pages, nc * sizeof *m->pages);
        m->frozen = realloc(m->frozen, nc);
        m->cap = nc;
    }
    m->pages[m->n] = calloc(1, BLCKSZ);
    m->frozen[m->n] = 0;
    return m->n++;
}
static const char *mem_read(void *r, BlockNumber b) { CryoMemRel *m = r; return b < m->n ? m->pages[b] : NULL; }
static char *mem_write(void *r, BlockNumber b) { CryoMemRel *m = r; return b < m->n ? m->pages[b] : NULL; }
static bool mem_frozen(void *r, BlockNumber b) { CryoMemRel *m = r; return b < m->n && m->frozen[b]; }
static const CryoRelOps mem_ops = {mem_nblocks, mem_read, mem_write, mem_extend, mem_frozen, NULL, NULL};

CryoMemRel *cryo_memrel_create(void)
{
    CryoMemRel *m = calloc(1, sizeof *m);
    if (m) {
        /* block 0 is the metapage (CRYO_META_PAGE, pg_cryogen.c:533-586) */
        BlockNumber b = mem_extend(m);
        CryoMetaPage *mp = (CryoMetaPage *)m->pages[b];
        mp->base.pd_upper = BLCKSZ;
        mp->base.pd_lower = sizeof(CryoMetaPage);
        mp->base.pd_special = BLCKSZ;
        mp->version = STORAGE_VERSION;
    }
    return m;
}
void cryo_memrel_destroy(CryoMemRel *m)
{
    BlockNumber i;
    if (!m) return;
    for (i = 0; i < m->n; i++) free(m->pages[i]);
    free(m->pages); free(m->frozen); free(m);
}
void cryo_memrel_bind(CryoMemRel *m, Oid relid, CryoRel *out) { out->relid = relid; out->handle = m; out->ops = &mem_ops; }
BlockNumber cryo_memrel_reserve(CryoMemRel *m) { return mem_extend(m); }
void cryo_memrel_set_frozen(CryoMemRel *m, BlockNumber b, bool f) { if (b < m->n) m->frozen[b] = f; }
const char *cryo_memrel_page(CryoMemRel *m, BlockNumber b) { return mem_read(m, b); }
BlockNumber cryo_memrel_nblocks(CryoMemRel *m) { return m->n; }
