/* syntax-check stand-in, see ../README */
#ifndef CRYO_STUB_LMGR_H
#define CRYO_STUB_LMGR_H
#include "utils/rel.h"
typedef int LOCKMODE;
#define ExclusiveLock 7
extern void LockRelationForExtension(Relation relation, LOCKMODE lockmode);
extern void UnlockRelationForExtension(Relation relation, LOCKMODE lockmode);
#endif
