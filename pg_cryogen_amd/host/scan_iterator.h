/*
 * scan_iterator.h -- block-range set of a sequential scan (reference scan_iterator.h:7-9,
 * scan_iterator.c:24-127), PostgreSQL-free.
 *
 * A cryo block is identified by the number of its first page; its continuation pages can lie
 * anywhere (another backend may have extended the relation in between, scan_iterator.c:4-15).
 * The iterator holds the set of page numbers not yet consumed as ordered [start,end] ranges,
 * starting with [1, InvalidBlockNumber): next() pops the lowest, exclude() removes a page that
 * turned out to be a continuation page (or that the cache already served).  The read-ahead
 * (cache.h: cryo_scan_next_batch) issues block starts in exactly this order.
 */
#ifndef CRYO_SCAN_ITERATOR_H
#define CRYO_SCAN_ITERATOR_H

#include "compression.h"

typedef struct SeqScanIterator SeqScanIterator;

SeqScanIterator *cryo_seqscan_iter_create(void);
void cryo_seqscan_iter_free(SeqScanIterator *iter);
BlockNumber cryo_seqscan_iter_next(SeqScanIterator *iter);
/* returns true if the block was in the set; with !miss_ok a miss raises the reference's
 * "block %u is not the part of seqscan iterator" error (scan_iterator.c:123-126) */
bool cryo_seqscan_iter_exclude(SeqScanIterator *iter, BlockNumber block, bool miss_ok);
/* addition: restart from block 1.  The reference's cryo_rescan (pg_cryogen.c:318-328) forgets to
 * do this, which its regression output pins as "LIMIT 3 returns 1 row" (expected/pg_cryogen.out:121-125) */
void cryo_seqscan_iter_reset(SeqScanIterator *iter);
int cryo_seqscan_iter_nranges(const SeqScanIterator *iter);
/* addition: a copy of the set, for looking ahead without consuming (cache.c: the read-ahead inside cryo_read_data_rel) */
SeqScanIterator *cryo_seqscan_iter_clone(const SeqScanIterator *iter);

#endif
