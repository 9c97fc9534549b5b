/* syntax-check stand-in, see ../README */
#ifndef CRYO_STUB_BUFPAGE_H
#define CRYO_STUB_BUFPAGE_H
extern void PageSetChecksumInplace(Page page, BlockNumber blkno);
#endif
