/* scan_iterator.c -- see scan_iterator.h */
#include "scan_iterator.h"

typedef struct { BlockNumber start, end; } BlockRange;

struct SeqScanIterator {
    BlockRange *r;
    int n, cap;
};

static void insert_range(SeqScanIterator *it, int pos, BlockNumber start, BlockNumber end)
{
    if (it->n == it->cap) {
        it->cap = it->cap ? it->cap * 2 : 8;
        it->r = realloc(it->r, (size_t)it->cap * sizeof *it->r);
    }
    memmove(it->r + pos + 1, it->r + pos, (size_t)(it->n - pos) * sizeof *it->r);
    it->r[pos].start = start;
    it->r[pos].end = end;
    it->n++;
}
static void delete_range(SeqScanIterator *it, int pos)
{
    memmove(it->r + pos, it->r + pos + 1, (size_t)(it->n - pos - 1) * sizeof *it->r);
    it->n--;
}

void cryo_seqscan_iter_reset(SeqScanIterator *it)
{
    it->n = 0;
    insert_range(it, 0, 1, InvalidBlockNumber); /* block 0 is the metapage */
}

SeqScanIterator *cryo_seqscan_iter_create(void)
{
    SeqScanIterator *it = calloc(1, sizeof *it);
    if (it) cryo_seqscan_iter_reset(it);
    return it;
}

void cryo_seqscan_iter_free(SeqScanIterator *it)
{
    if (it) { free(it->r); free(it); }
}

BlockNumber cryo_seqscan_iter_next(SeqScanIterator *it)
{
    BlockNumber res;
    if (it->n == 0) return InvalidBlockNumber;
    res = it->r[0].start++;
    if (it->r[0].start > it->r[0].end || it->r[0].start == 0 /* wrapped */) delete_range(it, 0);
    return res;
}

bool cryo_seqscan_iter_exclude(SeqScanIterator *it, BlockNumber block, bool miss_ok)
{
    int i;
    if (!it) return false;
    for (i = 0; i < it->n; i++) {
        BlockRange *r = &it->r[i];
        if (block >= r->start && block <= r->end) {
            if (block == r->start) r->start++;
            else if (block == r->end) r->end--;
            else {
                const BlockNumber old_end = r->end;
                r->end = block - 1;
                insert_range(it, i + 1, block + 1, old_end);
                return true;
            }
            if (r->start > r->end) delete_range(it, i);
            return true;
        }
    }
    if (!miss_ok)
        elog(ERROR, "pg_cryogen: iternal error; block %u is not the part of seqscan iterator", block);
    return false;
}

int cryo_seqscan_iter_nranges(const SeqScanIterator *it) { return it->n; }

SeqScanIterator *cryo_seqscan_iter_clone(const SeqScanIterator *it)
{
    SeqScanIterator *c = calloc(1, sizeof *c);
    if (!c) return NULL;
    c->cap = it->n > 8 ? it->n : 8;
    c->r = malloc((size_t)c->cap * sizeof *c->r);
    if (!c->r) { free(c); return NULL; }
    memcpy(c->r, it->r, (size_t)it->n * sizeof *c->r);
    c->n = it->n;
    return c;
}
