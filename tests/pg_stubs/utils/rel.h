/* syntax-check stand-in, see ../README */
#ifndef CRYO_STUB_REL_H
#define CRYO_STUB_REL_H
typedef struct RelationData *Relation;
extern Oid RelationGetRelid_stub(Relation rel);
#define RelationGetRelid(rel) RelationGetRelid_stub(rel)
#endif
