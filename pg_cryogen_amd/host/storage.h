/*
 * storage.h -- on-disk page formats and the in-block tuple layout of pg_cryogen
 * (reference storage.h:11-91), restated without PostgreSQL headers.  Sizes are part of the
 * storage contract and are checked by _Static_assert below:
 *   PageHeaderClone 24, CryoMetaPage 40, CryoPageHeader 32, CryoFirstPageHeader 48.
 */
#ifndef __STORAGE_H__
#define __STORAGE_H__

#include "compression.h"

#define STORAGE_VERSION 1
#define CRYO_META_PAGE 0
#define CRYO_BLCKSZ cryo_blcksz /* run-time; default 1 MiB as reference storage.h:18 */

typedef struct
{
    uint64 pd_lsn; /* PageXLogRecPtr {xlogid, xrecoff} */
    uint16 pd_checksum;
    uint16 pd_flags;
    uint16 pd_lower;
    uint16 pd_upper;
    uint16 pd_special;
    uint16 pd_pagesize_version;
    TransactionId pd_prune_xid;
} PageHeaderClone;

typedef struct
{
    PageHeaderClone base;
    uint16 version;
    uint64 ntuples;
} CryoMetaPage;

typedef struct
{
    PageHeaderClone base;
    BlockNumber first;
    BlockNumber next;
} CryoPageHeader;

typedef struct
{
    CryoPageHeader cryo_base;
    TransactionId created_xid;
    CompressionMethod compression_method;
    uint32 compressed_size;
    uint16 npages;
} CryoFirstPageHeader;

_Static_assert(sizeof(PageHeaderClone) == 24, "PageHeaderData layout");
_Static_assert(sizeof(CryoMetaPage) == 40, "CryoMetaPage layout");
_Static_assert(sizeof(CryoPageHeader) == 32, "CryoPageHeader layout");
_Static_assert(sizeof(CryoFirstPageHeader) == 48, "CryoFirstPageHeader layout");

#define CryoPageHeaderSize(page, block) \
    ((page)->first == (block) ? sizeof(CryoFirstPageHeader) : sizeof(CryoPageHeader))

typedef struct
{
    uint32 off;
    uint32 len;
} CryoItemId;

typedef struct
{
    uint32 lower;
    uint32 upper;
    char data[];
} CryoDataHeader;

#define CryoDataHeaderSize offsetof(CryoDataHeader, data)

void cryo_init_page(CryoDataHeader *hdr);
int cryo_storage_insert(CryoDataHeader *d, HeapTuple tuple);
HeapTuple cryo_storage_fetch(CryoDataHeader *d, int pos, HeapTuple tuple);
int cryo_storage_ntuples(const CryoDataHeader *d);

#endif /* __STORAGE_H__ */
