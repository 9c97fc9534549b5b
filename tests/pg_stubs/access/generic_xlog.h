/* syntax-check stand-in, see ../README */
#ifndef CRYO_STUB_GENERIC_XLOG_H
#define CRYO_STUB_GENERIC_XLOG_H
#include "storage/bufmgr.h"
#define GENERIC_XLOG_FULL_IMAGE 0x0001
typedef struct GenericXLogState GenericXLogState;
extern GenericXLogState *GenericXLogStart(Relation relation);
extern Page GenericXLogRegisterBuffer(GenericXLogState *state, Buffer buffer, int flags);
extern uint64 GenericXLogFinish(GenericXLogState *state);
extern void GenericXLogAbort(GenericXLogState *state);
#endif
