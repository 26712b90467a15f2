/*
 * syzgy_pager.h -- C ABI of the spanfile pager: reads a SyzgyDB collection file
 * (.dat) into the dense rows x row_bytes matrix + row -> document-id table the
 * scan (syzgy_scan.h) works on, without the Go runtime.
 *
 * It restates the READ side of the reference's storage layer, nothing else:
 *   spanfile.go:1-22     span grammar (magic, u32 length, 7-code sequence number,
 *                        record id, streams, padding, CRC32-IEEE trailer)
 *   spanfile.go:282-357  scanFile: walk spans from offset 0; magic 0 = rest is
 *                        free; a span with a bad checksum or parse error is
 *                        skipped by its length; FREE spans are skipped; for a
 *                        record id seen twice the HIGHEST sequence number wins
 *   spanfile.go:568-661  7-code varints (MSB-first base-128)
 *   spanfile.go:730-849  parseSpan / verifyChecksum
 *   collection.go:241-272 the header record "" holds the CollectionOptions JSON,
 *                        :446-450 stream 0 = metadata, stream 1 = packed vector
 *   spanfile.go:540-560  rows come out in IterateSortedRecords order
 *                        (sort.Strings over the decimal record ids)
 * The Go binding does not need this (it has the spanfile in memory); it is the
 * first "next" row of SURVEY.md 8f: C/C++/Python hosts and the GPU box can open
 * real collection files.  Pure host code, no device work.
 */
#ifndef SYZGY_PAGER_H
#define SYZGY_PAGER_H

#include <stdint.h>
#include "syzgy_scan.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct szg_pager szg_pager;

#define SZG_E_IO (-8)      /* cannot open / map the file */
#define SZG_E_FORMAT (-9)  /* no usable header record / inconsistent vector sizes */

/* Open and scan a collection file (read-only mmap). n_threads <= 0: one per core (CRC pass). */
int szg_pager_open(szg_pager **out, const char *path, int n_threads);
void szg_pager_close(szg_pager *p);

/* CollectionOptions of the header record (collection.go:31-48). */
int szg_pager_options(const szg_pager *p, int *dim, int *quant_bits, int *metric);

/* Live records with a numeric id and a vector stream of the right size. */
uint64_t szg_pager_count(const szg_pager *p);
/* Spans skipped during the scan: bad checksum or unparsable (spanfile.go:313-327). */
uint64_t szg_pager_skipped(const szg_pager *p);

/* Document ids in visit order (row r -> ids[r]). */
int szg_pager_ids(const szg_pager *p, uint64_t *ids);
/* count x row_bytes packed vectors in visit order, reference element encoding. */
int szg_pager_vectors(const szg_pager *p, uint8_t *out, uint64_t capacity_bytes);
/* Metadata (stream 0) of row r: pointer into the mapping, valid until close. */
int szg_pager_metadata(const szg_pager *p, uint64_t row, const uint8_t **data, uint64_t *len);

/* Page every vector into a scan handle created with the same options (= szg_index_load). */
int szg_pager_load(const szg_pager *p, szg_index *ix);

#ifdef __cplusplus
}
#endif
#endif
