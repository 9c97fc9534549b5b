/*
 * storage.c -- the uncompressed payload the codec sees: the in-block tuple layout
 * (reference storage.c:15-68).  [0,4) lower, [4,8) upper, CryoItemId{off,len} array upward
 * from byte 8 (1-based positions), MAXALIGNed tuples downward from CRYO_BLCKSZ, zero gap
 * between; a block is full when the tuple plus its item id no longer fits or after
 * MaxHeapTuplesPerPage - 1 = 290 tuples (storage.c:10,32-33).
 */
#include "storage.h"

#define MAX_TUPLES_PER_PAGE MaxHeapTuplesPerPage

static inline int last_item_pos(const CryoDataHeader *d)
{
    return (int)((d->lower - CryoDataHeaderSize) / sizeof(CryoItemId));
}

void cryo_init_page(CryoDataHeader *hdr)
{
    memset(hdr, 0, CRYO_BLCKSZ);
    hdr->lower = CryoDataHeaderSize;
    hdr->upper = (uint32)CRYO_BLCKSZ;
}

/* returns the 1-based item position, or -1 when the block is full */
int cryo_storage_insert(CryoDataHeader *d, HeapTuple tuple)
{
    CryoItemId item;

    if ((tuple->t_len + sizeof(CryoItemId)) > (d->upper - d->lower)
        || last_item_pos(d) + 1 >= MAX_TUPLES_PER_PAGE)
        return -1;

    d->upper -= (uint32)MAXALIGN(tuple->t_len);
    memcpy((char *)d + d->upper, tuple->t_data, tuple->t_len);

    item.off = d->upper;
    item.len = tuple->t_len;
    memcpy((char *)d + d->lower, &item, sizeof item);
    d->lower += sizeof(CryoItemId);

    return last_item_pos(d);
}

HeapTuple cryo_storage_fetch(CryoDataHeader *d, int pos, HeapTuple tuple)
{
    CryoItemId *item = (CryoItemId *)d->data + pos - 1; /* pos is 1-based */
    tuple->t_data = (char *)d + item->off;
    tuple->t_len = item->len;
    return tuple;
}

int cryo_storage_ntuples(const CryoDataHeader *d) { return last_item_pos(d); }
