/*
 * pg_compat.h -- the handful of PostgreSQL names the host side uses, for builds WITHOUT
 * PostgreSQL headers (this image has no pg_config / postgres.h).  With -DCRYO_HAVE_POSTGRES
 * the real headers are used instead and this file is not included.
 *
 * Only names; no PostgreSQL behaviour is emulated beyond malloc-backed palloc and an
 * overridable elog(ERROR) (PostgreSQL longjmps out of elog(ERROR); here the default handler
 * aborts, tests install their own with cryo_compat_set_error_handler()).
 */
#ifndef CRYO_PG_COMPAT_H
#define CRYO_PG_COMPAT_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef size_t Size;
typedef uint8_t uint8;
typedef uint16_t uint16;
typedef uint32_t uint32;
typedef uint64_t uint64;
typedef uint32 BlockNumber;
typedef uint32 TransactionId;
typedef unsigned int Oid;
#define InvalidOid ((Oid)0)

#define InvalidBlockNumber ((BlockNumber)0xFFFFFFFF)
#define BlockNumberIsValid(b) ((BlockNumber)(b) != InvalidBlockNumber)
#define FrozenTransactionId ((TransactionId)2)
#define BLCKSZ 8192
#define MAXALIGN(x) (((size_t)(x) + 7u) & ~(size_t)7u)
#define MaxHeapTuplesPerPage 291 /* (BLCKSZ - 24) / (MAXALIGN(23) + 4) at 8 KiB pages */

#define ERROR 20
#define DEBUG1 14

#ifdef __cplusplus
extern "C" {
#endif
typedef void (*cryo_error_handler)(int elevel, const char *msg);
void cryo_compat_set_error_handler(cryo_error_handler h);
void cryo_compat_elog(int elevel, const char *fmt, ...);
#ifdef __cplusplus
}
#endif

#define elog(level, ...) cryo_compat_elog((level), __VA_ARGS__)
#define palloc(sz) malloc(sz)
#define pfree(p) free(p)
#define Assert(x) ((void)0)

/* minimal heap-tuple view used by the block layout (htup.h: t_len, t_data) */
typedef struct HeapTupleData {
    uint32 t_len;
    void *t_data;
} HeapTupleData;
typedef HeapTupleData *HeapTuple;

#endif
