/*
 * cryo_pg_rel.h -- the bufmgr-backed binding of a PostgreSQL Relation to the page-access callbacks of this repo's
 * staging / cache code (pg_cryogen_amd/host/staging.h: CryoRelOps).  PGXS builds only (pg/Makefile BATCH=1).
 */
#ifndef CRYO_PG_REL_H
#define CRYO_PG_REL_H
#ifdef CRYO_HAVE_POSTGRES
#include "postgres.h"

#include "access/generic_xlog.h"
#include "storage/bufmgr.h"
#include "utils/rel.h"

#include "cache.h"

#define CRYO_PG_PINS 8 /* pages a chain walk has pinned at once: the current one, briefly the next */

typedef struct CryoPgRel
{
    Relation    rel;
    /* read side: pins taken by read_page, dropped by release_page */
    BlockNumber pin_block[CRYO_PG_PINS];
    Buffer      pin_buf[CRYO_PG_PINS];
    /* write side: the page being filled */
    Buffer      wbuf;
    GenericXLogState *xlog;
} CryoPgRel;

/* bind an open relation; `store` is caller memory that lives as long as the CryoRel is used */
void cryo_pg_bind(Relation rel, CryoPgRel *store, CryoRel *out);

#endif /* CRYO_HAVE_POSTGRES */
#endif /* CRYO_PG_REL_H */
