/* syntax-check stand-in, see README in this directory */
#ifndef CRYO_STUB_POSTGRES_H
#define CRYO_STUB_POSTGRES_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h> /* c.h pulls these in for every PostgreSQL source */
#include <stdio.h>
#include <stdarg.h>
typedef size_t Size;
typedef uint8_t uint8;
typedef uint16_t uint16;
typedef uint32_t uint32;
typedef uint64_t uint64;
typedef uint32 BlockNumber;
typedef uint32 TransactionId;
typedef unsigned int Oid;
#define InvalidOid ((Oid)0)
typedef char *Pointer;
typedef Pointer Page;
#define InvalidBlockNumber ((BlockNumber)0xFFFFFFFF)
#define BlockNumberIsValid(b) ((BlockNumber)(b) != InvalidBlockNumber)
#define FrozenTransactionId ((TransactionId)2)
#define BLCKSZ 8192
#define MAXALIGN(x) (((size_t)(x) + 7u) & ~(size_t)7u)
#define MaxHeapTuplesPerPage 291
#define ERROR 20
#define DEBUG1 14
extern void elog_stub(int elevel, const char *fmt, ...);
#define elog(level, ...) elog_stub((level), __VA_ARGS__)
extern void *palloc(Size size);
extern void pfree(void *p);
#define Assert(x) ((void)(x))
typedef struct HeapTupleData { uint32 t_len; void *t_data; } HeapTupleData;
typedef HeapTupleData *HeapTuple;
#endif
