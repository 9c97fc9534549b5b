/*
 * cache.h -- decompressed-block cache (reference cache.h:25-35, cache.c:17-401).
 *
 * Same API and contract as the reference: key (relid, first block number); a slot is free,
 * loaded or pinned; a pinned slot is the insert buffer of a modify state and is never
 * evicted; cryo_read_data loads a missing block (chain walk + decompress); CACHE_IS_FULL only
 * when every slot is pinned.  Deliberate fixes: true LRU eviction (the
 * reference's scan never updates min_ts and evicts the last unpinned slot, cache.c:194-207),
 * the per-slot block list is sized for the largest possible chain (cache.c:44 overruns by up
 * to two entries), the reassembly buffer is freed (cache.c:134 leaks it), configurable slot
 * count (CACHE_SIZE 16, cache.c:17) and CRYO_ERR_EMPTY_BLOCK has a message.
 * Addition: cryo_read_data_batch = read-ahead of K blocks decoded by ONE codec call.
 */
#ifndef __CACHE_H__
#define __CACHE_H__

#include "staging.h"

#define InvalidCacheEntry -1
typedef int CacheEntry;

void cryo_init_cache(void);                 /* 16 slots of cryo_blcksz bytes */
int cryo_cache_configure(int nslots);       /* (re)allocate; drops all content */
void cryo_cache_shutdown(void);

/* The cache works on a CryoRel (staging.h: relation oid + page access callbacks).  The reference's entry points take
 * a PostgreSQL Relation (cache.h:25-27); in a PGXS build (-DCRYO_HAVE_POSTGRES) those exact prototypes are exported by
 * pg/cryo_pg_rel.c as wrappers that bind the Relation to the bufmgr-backed callbacks, so that the seven call sites of
 * reference pg_cryogen.c (:117,265,389,423,873,1017 and _PG_init :172) compile unchanged.  Without PostgreSQL the same
 * names take the CryoRel directly (the PG-free harness of the tests). */
CryoError cryo_read_data_rel(CryoRel *rel, void *iter, BlockNumber block, CacheEntry *result);
CacheEntry cryo_cache_allocate_rel(CryoRel *rel, BlockNumber blockno);
#ifdef CRYO_HAVE_POSTGRES
#include "utils/relcache.h"
#include "scan_iterator.h"
CryoError cryo_read_data(Relation rel, SeqScanIterator *iter, BlockNumber block, CacheEntry *result);
CacheEntry cryo_cache_allocate(Relation rel, BlockNumber blockno);
#else
CryoError cryo_read_data(CryoRel *rel, void *iter, BlockNumber block, CacheEntry *result);
CacheEntry cryo_cache_allocate(CryoRel *rel, BlockNumber blockno);
#endif
CryoError cryo_read_data_batch(CryoRel *rel, const BlockNumber *blocks, int k, CacheEntry *results,
                               CryoError *errors);
/* read-ahead in seq-scan order (iter = SeqScanIterator*, scan_iterator.h); returns blocks delivered */
int cryo_scan_next_batch(CryoRel *rel, void *iter, int k, BlockNumber *starts, CacheEntry *entries,
                         CryoError *errors);
void cryo_cache_release(CacheEntry entry);
void cryo_cache_invalidate_relation(Oid relid);

uint32 cryo_cache_get_pg_nblocks(CacheEntry entry);
char *cryo_cache_get_data(CacheEntry entry);
TransactionId cryo_cache_get_xid(CacheEntry entry);
const char *cryo_cache_err(CryoError err);

/* statistics for tests */
uint64 cryo_cache_hits(void);
uint64 cryo_cache_misses(void);
uint64 cryo_cache_codec_calls(void);

#endif /* __CACHE_H__ */
