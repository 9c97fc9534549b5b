/* declaration-only stand-in (see README) */
#ifndef STUB_RELCACHE_H
#define STUB_RELCACHE_H
typedef struct RelationData *Relation;
#endif
