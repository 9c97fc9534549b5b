/* syntax-check stand-in, see ../README */
#ifndef CRYO_STUB_VISIBILITYMAP_H
#define CRYO_STUB_VISIBILITYMAP_H
#include "storage/bufmgr.h"
#define VISIBILITYMAP_ALL_VISIBLE 0x01
#define VISIBILITYMAP_ALL_FROZEN 0x02
extern uint8 visibilitymap_get_status(Relation rel, BlockNumber heapBlk, Buffer *vmbuf);
#endif
