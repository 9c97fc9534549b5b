/* syntax-check stand-in, see ../README */
#ifndef CRYO_STUB_BUFMGR_H
#define CRYO_STUB_BUFMGR_H
#include "utils/rel.h"
typedef int Buffer;
#define InvalidBuffer 0
#define BufferIsValid(b) ((b) != InvalidBuffer)
#define P_NEW InvalidBlockNumber
#define BUFFER_LOCK_UNLOCK 0
#define BUFFER_LOCK_SHARE 1
#define BUFFER_LOCK_EXCLUSIVE 2
extern Buffer ReadBuffer(Relation reln, BlockNumber blockNum);
extern void ReleaseBuffer(Buffer buffer);
extern void UnlockReleaseBuffer(Buffer buffer);
extern void MarkBufferDirty(Buffer buffer);
extern void LockBuffer(Buffer buffer, int mode);
extern BlockNumber BufferGetBlockNumber(Buffer buffer);
extern Page BufferGetPage(Buffer buffer);
extern BlockNumber RelationGetNumberOfBlocks_stub(Relation relation);
#define RelationGetNumberOfBlocks(reln) RelationGetNumberOfBlocks_stub(reln)
#endif
