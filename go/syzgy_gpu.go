//go:build syzgy_gpu

// Package syzgydb: cgo binding of libsyzgy_scan.so (include/syzgy_scan.h), the
// MI355X-native exact scan behind Collection.Search{Precision:"exact"}.
//
// This file is what a SyzgyDB maintainer adds to the reference tree
// (github.com/smhanov/syzgydb, package syzgydb); INTEGRATION.md shows the few
// lines of collection.go that call it.  It cannot be compiled in this
// repository's image (no Go toolchain), so it is kept deliberately thin: all
// logic lives behind the C ABI, where the C++/ctypes drivers exercise it.
//
// Build:  CGO_CFLAGS="-I$REPO/include" CGO_LDFLAGS="-L$REPO/syzgydb_amd -lsyzgy_scan" \
//         go build -tags syzgy_gpu ./...
package syzgydb

/*
#cgo LDFLAGS: -lsyzgy_scan
#include <stdlib.h>
#include "syzgy_scan.h"
*/
import "C"

import (
	"container/heap"
	"fmt"
	"math"
	"os"
	"sort"
	"strconv"
	"sync"
	"unsafe"
)

// gpuMirror is the HBM mirror of one Collection's packed vectors plus the
// row -> document id table (rows are in IterateSortedRecords order, i.e. the
// reference's deterministic visit order, spanfile.go:540-560).
type gpuMirror struct {
	h     *C.szg_index
	ids   []uint64          // row -> document id
	rowOf map[uint64]uint64 // document id -> row
	dirty bool              // set by mutators that could not be applied incrementally
	// Search runs under the collection's RLock only, so searches are concurrent with each
	// other; a reload (which rewrites ids / rowOf and calls szg_index_load, both of which
	// need exclusive access) takes mu for writing, every search takes it for reading.
	mu      sync.RWMutex
	lastID  string // greatest decimal id string among the loaded rows (visit order = sort.Strings)
	// Row order only matters where the reference's own answer depends on its visit order: ties at the k
	// boundary, equal distances in a radius result (collection.go:608, :536-564).  Production iterates a Go map
	// (random order per run, spanfile.go:525), so there every order is a reference order and rows are simply
	// appended.  Only the reference's deterministic mode (myRandom.rand != nil: seeded tests, spanfile.go:522)
	// pins the order to sort.Strings of the ids; there an out-of-order append sets orderStale, and the mirror is
	// re-paged in order the first time an answer actually contains such a tie -- not on every add.
	orderStale bool
	version uint64 // bumped by add / remove: filter verdicts are cached per (key, version)
	masks   map[string]cachedMask
}

// cachedMask is one filter's verdicts, one bit per row, valid for one collection version.
type cachedMask struct {
	version uint64
	bits    []C.uint64_t
}

// newGPUMirror pages every record's stream 1 (the packed vector, exactly the
// bytes encodeDocument wrote, collection.go:713-743) into HBM.  Called from
// NewCollection where the LSH tree is rebuilt (collection.go:297-311).
func newGPUMirror(c *Collection, devices []int) (*gpuMirror, error) {
	m := &gpuMirror{rowOf: map[uint64]uint64{}, masks: map[string]cachedMask{}}
	var devp *C.int
	cdev := make([]C.int, len(devices))
	for i, d := range devices {
		cdev[i] = C.int(d)
	}
	if len(cdev) > 0 {
		devp = &cdev[0]
	}
	rc := C.szg_index_create(&m.h, C.int(c.DimensionCount), C.int(c.Quantization),
		C.int(c.DistanceMethod), devp, C.int(len(cdev)))
	if rc != C.SZG_OK {
		return nil, fmt.Errorf("szg_index_create: %s (%s)", C.GoString(C.szg_strerror(rc)),
			C.GoString(C.szg_last_error()))
	}
	// Optional: float32 collections can answer lone Searches from an 8-bit sketch of the rows (a quarter of
	// the bytes per query, +25 % device memory, same answers: DESIGN.md 4.5).  Off unless asked for.
	if os.Getenv("SYZGY_GPU_SKETCH") == "1" && c.Quantization == 32 {
		name := C.CString("sketch")
		C.szg_set_option(m.h, name, 1)
		C.free(unsafe.Pointer(name))
	}
	if err := m.reload(c); err != nil {
		m.close()
		return nil, err
	}
	return m, nil
}

// reload copies all vectors into one contiguous host buffer and loads it.
func (m *gpuMirror) reload(c *Collection) error {
	rowBytes := getVectorSize(c.Quantization, c.DimensionCount)
	var recordIDs []string
	for id := range c.spanfile.index {
		if id != "" {
			recordIDs = append(recordIDs, id)
		}
	}
	sort.Strings(recordIDs) // IterateSortedRecords order
	buf := make([]byte, 0, len(recordIDs)*rowBytes)
	m.ids = m.ids[:0]
	m.rowOf = map[uint64]uint64{}
	m.lastID = ""
	m.version++ // rows are renumbered: cached filter masks no longer apply
	for _, rid := range recordIDs {
		id, err := strconv.ParseUint(rid, 10, 64)
		if err != nil {
			continue // collection.go:676-678 skips non-numeric record ids
		}
		span, err := c.spanfile.ReadRecord(rid)
		if err != nil {
			return err
		}
		m.rowOf[id] = uint64(len(m.ids))
		m.ids = append(m.ids, id)
		m.lastID = rid
		buf = append(buf, span.DataStreams[1].Data...)
	}
	var p *C.uint8_t
	if len(buf) > 0 {
		p = (*C.uint8_t)(unsafe.Pointer(&buf[0]))
	}
	if rc := C.szg_index_load(m.h, p, C.uint64_t(len(m.ids))); rc != C.SZG_OK {
		return fmt.Errorf("szg_index_load: %s", C.GoString(C.szg_last_error()))
	}
	m.dirty = false
	m.orderStale = false
	return nil
}

// add mirrors AddDocument (collection.go:427-457); called under c.mutex.Lock, so no search
// is in flight.  A new row is always appended in place (one small H2D copy); see orderStale for
// what the reference's deterministic visit order asks for on top of that.
func (m *gpuMirror) add(id uint64, encoded []byte) {
	m.version++
	p := (*C.uint8_t)(unsafe.Pointer(&encoded[0]))
	if row, ok := m.rowOf[id]; ok {
		if C.szg_index_overwrite(m.h, C.uint64_t(row), p) != C.SZG_OK {
			m.dirty = true
		}
		return
	}
	if m.dirty {
		return // a failed call left the mirror behind the file: the next search reloads everything
	}
	if C.szg_index_append(m.h, p, 1) != C.SZG_OK {
		m.dirty = true
		return
	}
	m.rowOf[id] = uint64(len(m.ids))
	m.ids = append(m.ids, id)
	rid := strconv.FormatUint(id, 10)
	if rid < m.lastID { // decimal STRING order: "10" < "9"
		if myRandom.rand != nil {
			m.orderStale = true
		}
	} else {
		m.lastID = rid
	}
}

// addBulk is AddDocument for a block of new documents (bulk ingest): the float64 vectors are quantized and
// packed on the device exactly as encodeDocument / quantize do (szg_index_append_f64; collection.go:713-743,
// quantization.go:5-23), one call for the block.  ids must be new and distinct; called under c.mutex.Lock after
// the records have been written to the spanfile.
func (m *gpuMirror) addBulk(ids []uint64, vectors []float64, dim int) bool {
	if m.dirty || len(ids) == 0 || len(vectors) != len(ids)*dim {
		return false
	}
	for _, id := range ids {
		if _, ok := m.rowOf[id]; ok {
			return false
		}
	}
	m.version++
	if C.szg_index_append_f64(m.h, (*C.double)(unsafe.Pointer(&vectors[0])), C.uint64_t(len(ids))) != C.SZG_OK {
		m.dirty = true
		return false
	}
	for _, id := range ids {
		m.rowOf[id] = uint64(len(m.ids))
		m.ids = append(m.ids, id)
		rid := strconv.FormatUint(id, 10)
		if rid < m.lastID {
			if myRandom.rand != nil {
				m.orderStale = true
			}
		} else {
			m.lastID = rid
		}
	}
	return true
}

// remove mirrors removeDocument (collection.go:511-521); called under c.mutex.Lock.
func (m *gpuMirror) remove(id uint64) {
	m.version++
	if row, ok := m.rowOf[id]; ok {
		if C.szg_index_tombstone(m.h, C.uint64_t(row)) != C.SZG_OK {
			m.dirty = true
		}
		delete(m.rowOf, id)
	}
}

// touch mirrors UpdateDocument (collection.go:490-509): the vectors do not change, but filters
// see metadata, so cached filter masks no longer apply.  Called under c.mutex.Lock.
func (m *gpuMirror) touch() { m.version++ }

func (m *gpuMirror) close() {
	if m.h != nil {
		C.szg_index_destroy(m.h)
		m.h = nil
	}
}

// allowBits evaluates args.Filter over every live row (collection.go:592-594);
// a Go closure cannot run on the GPU, its verdicts travel as one bit per row.
func (m *gpuMirror) allowBits(c *Collection, filter FilterFn) []C.uint64_t {
	words := (len(m.ids) + 63) / 64
	bits := make([]C.uint64_t, words)
	for id, row := range m.rowOf {
		span, err := c.spanfile.ReadRecord(fmt.Sprintf("%d", id))
		if err != nil {
			continue
		}
		if filter(id, span.DataStreams[0].Data) {
			bits[row/64] |= 1 << (row % 64)
		}
	}
	return bits
}

// allowBitsKeyed is allowBits behind a cache: a filter's verdicts only change when the
// collection does, so they are kept per (key, version) -- the REST layer's natural key is the
// filter text it compiled (rest.go:429-436).  key == "" evaluates every time (a bare closure
// has no identity to cache by).  Called with m.mu read-locked; the cache has its own lock.
var maskCacheMu sync.Mutex

func (m *gpuMirror) allowBitsKeyed(c *Collection, filter FilterFn, key string) []C.uint64_t {
	if key == "" {
		return m.allowBits(c, filter)
	}
	maskCacheMu.Lock()
	if hit, ok := m.masks[key]; ok && hit.version == m.version {
		maskCacheMu.Unlock()
		return hit.bits
	}
	maskCacheMu.Unlock()
	bits := m.allowBits(c, filter)
	maskCacheMu.Lock()
	if len(m.masks) > 32 {
		m.masks = map[string]cachedMask{}
	}
	m.masks[key] = cachedMask{version: m.version, bits: bits}
	maskCacheMu.Unlock()
	return bits
}

// searchExact replaces the hot loop of Collection.Search (collection.go:672-684
// plus the pop loop :694-697).  ok == false tells the caller to run the
// reference's own CPU loop (keeps Search's no-error signature).
func (m *gpuMirror) searchExact(c *Collection, args SearchArgs) (results []SearchResult, ok bool) {
	return m.searchExactKeyed(c, args, "")
}

// ensureFresh reloads the mirror if a mutator could not be applied (dirty).  The caller holds only
// c.mutex.RLock: other searches may be inside the library right now.  The mirror's own lock makes the
// reload exclusive; whoever gets it first reloads.
func (m *gpuMirror) ensureFresh(c *Collection, alsoOrder bool) bool {
	m.mu.RLock()
	stale := m.dirty || (alsoOrder && m.orderStale)
	m.mu.RUnlock()
	if !stale {
		return true
	}
	m.mu.Lock()
	defer m.mu.Unlock()
	if m.dirty || (alsoOrder && m.orderStale) {
		if err := m.reload(c); err != nil {
			return false
		}
	}
	return true
}

// fullReplays reads the library's count of queries it re-answered by the exact replay in ROW order (equal
// distances or NaN among the best k+1: the reference's answer depends on its visit order there).
func (m *gpuMirror) fullReplays() uint64 {
	var st C.szg_stats
	if C.szg_get_stats(m.h, &st) != C.SZG_OK {
		return 0
	}
	return uint64(st.full_replays)
}

// searchExactKeyed: filterKey names args.Filter for the bitmask cache ("" = do not cache).
func (m *gpuMirror) searchExactKeyed(c *Collection, args SearchArgs, filterKey string) (results []SearchResult, ok bool) {
	if !m.ensureFresh(c, false) {
		return nil, false
	}
	m.mu.RLock()
	orderStale := m.orderStale
	before := uint64(0)
	if orderStale {
		before = m.fullReplays()
	}
	rows, dist, ok := m.searchRows(c, args, filterKey)
	tie := false
	if ok && orderStale {
		// deterministic mode with rows out of sort.Strings order: did this answer depend on the visit order?
		// (a concurrent search's replay can only make this fire needlessly, never hide a tie.  The counter only
		// moves with the library's tie_mode 0 -- the default, which this binding never changes; a host that sets
		// tie_mode 1 must re-page up front instead, as syzgydb_amd/collection.py does)
		tie = m.fullReplays() != before
		if args.Radius > 0 {
			seen := make(map[float64]bool, len(dist))
			for _, d := range dist {
				if seen[float64(d)] {
					tie = true
				}
				seen[float64(d)] = true
			}
		}
	}
	if ok && !tie {
		results, ok = m.resultsOf(c, rows, dist)
	}
	m.mu.RUnlock()
	if !ok || !tie {
		return results, ok
	}
	if !m.ensureFresh(c, true) { // re-page in sort.Strings order, then answer again
		return nil, false
	}
	m.mu.RLock()
	defer m.mu.RUnlock()
	rows, dist, ok = m.searchRows(c, args, filterKey)
	if !ok {
		return nil, false
	}
	return m.resultsOf(c, rows, dist)
}

// searchRows runs one exact search in the library: rows (mirror order) and the reference's float64 distances.
// Called with m.mu read-locked.
func (m *gpuMirror) searchRows(c *Collection, args SearchArgs, filterKey string) (rows []C.uint64_t, dist []C.double, ok bool) {
	if len(m.ids) == 0 {
		return nil, nil, true // empty collection: no results (collection_test.go:294-309)
	}
	if len(args.Vector) != c.DimensionCount {
		return nil, nil, false
	}
	var allow *C.uint64_t
	var keep []C.uint64_t
	if args.Filter != nil {
		keep = m.allowBitsKeyed(c, args.Filter, filterKey)
		allow = &keep[0]
	}
	q := (*C.double)(unsafe.Pointer(&args.Vector[0]))
	if args.Radius > 0 { // K is ignored (collection.go:598-605)
		capacity := 1 << 16 // a truncated call costs a second sweep
		for {
			rows = make([]C.uint64_t, capacity)
			dist = make([]C.double, capacity)
			var total C.uint64_t
			rc := C.szg_search_radius(m.h, q, C.double(args.Radius), allow, &rows[0], &dist[0],
				C.uint64_t(capacity), &total)
			if rc == C.SZG_E_TRUNCATED {
				capacity = int(total)
				continue
			}
			if rc != C.SZG_OK {
				return nil, nil, false
			}
			return rows[:int(total)], dist[:int(total)], true
		}
	}
	rows = make([]C.uint64_t, args.K)
	dist = make([]C.double, args.K)
	var count C.int32_t
	rc := C.szg_search_topk(m.h, q, 1, C.int(args.K), allow, &rows[0], &dist[0], &count)
	_ = keep
	if rc != C.SZG_OK {
		return nil, nil, false
	}
	return rows[:int(count)], dist[:int(count)], true
}

// resultsOf turns rows into SearchResults: row -> document id, metadata = the live mmap slice, as in
// getDocument (collection.go:476); the caller still holds c.mutex.RLock.  Called with m.mu read-locked.
func (m *gpuMirror) resultsOf(c *Collection, rows []C.uint64_t, dist []C.double) ([]SearchResult, bool) {
	results := make([]SearchResult, len(rows)) // [] not null when empty
	for i := range rows {
		id := m.ids[rows[i]]
		span, err := c.spanfile.ReadRecord(fmt.Sprintf("%d", id))
		if err != nil {
			return nil, false
		}
		results[i] = SearchResult{ID: id, Metadata: span.DataStreams[0].Data, Distance: float64(dist[i])}
	}
	return results, true
}

// searchExactBatch answers a list of exact top-k Searches with the same K in ONE library call: batches of two or
// more share a sweep on the matrix cores (up to 96 queries per pass of the corpus), which is how a caller that holds
// many queries -- a bulk endpoint beside rest.go:371-487, an offline job -- reaches that path directly instead of
// through coalesced goroutines.  Each query keeps its own Filter (filterKeys optional, parallel to args).  Radius
// searches (K ignored, collection.go:598-605) go through szg_search_radius_batch with their own radii.  ok ==
// false: run the Searches one by one.
func (m *gpuMirror) searchExactBatch(c *Collection, args []SearchArgs, filterKeys []string) (results [][]SearchResult, ok bool) {
	if len(args) == 0 {
		return nil, true
	}
	radius := args[0].Radius > 0
	for _, a := range args {
		if len(a.Vector) != c.DimensionCount || (a.Radius > 0) != radius || (!radius && a.K != args[0].K) {
			return nil, false
		}
	}
	if !m.ensureFresh(c, true) { // (a batch is not worth the tie bookkeeping: deterministic mode re-pages first)
		return nil, false
	}
	m.mu.RLock()
	defer m.mu.RUnlock()
	nq, dim := len(args), c.DimensionCount
	results = make([][]SearchResult, nq)
	if len(m.ids) == 0 {
		for i := range results {
			results[i] = make([]SearchResult, 0)
		}
		return results, true
	}
	q := make([]float64, nq*dim)
	words := (len(m.ids) + 63) / 64
	var masks []C.uint64_t
	anyFilter := false
	for _, a := range args {
		anyFilter = anyFilter || a.Filter != nil
	}
	if anyFilter {
		masks = make([]C.uint64_t, nq*words)
	}
	for i, a := range args {
		copy(q[i*dim:(i+1)*dim], a.Vector)
		if !anyFilter {
			continue
		}
		if a.Filter == nil {
			for w := 0; w < words; w++ {
				masks[i*words+w] = ^C.uint64_t(0)
			}
			continue
		}
		key := ""
		if i < len(filterKeys) {
			key = filterKeys[i]
		}
		copy(masks[i*words:(i+1)*words], m.allowBitsKeyed(c, a.Filter, key))
	}
	var allow *C.uint64_t
	if anyFilter {
		allow = &masks[0]
	}
	qp := (*C.double)(unsafe.Pointer(&q[0]))
	if radius {
		radii := make([]C.double, nq)
		for i, a := range args {
			radii[i] = C.double(a.Radius)
		}
		off := make([]C.uint64_t, nq+1)
		capacity := 1 << 16
		var rows []C.uint64_t
		var dist []C.double
		for {
			rows = make([]C.uint64_t, capacity)
			dist = make([]C.double, capacity)
			rc := C.szg_search_radius_batch(m.h, qp, C.int(nq), &radii[0], allow, &rows[0], &dist[0],
				C.uint64_t(capacity), &off[0])
			if rc == C.SZG_E_TRUNCATED {
				capacity = int(off[nq])
				continue
			}
			if rc != C.SZG_OK {
				return nil, false
			}
			break
		}
		for i := 0; i < nq; i++ {
			if results[i], ok = m.resultsOf(c, rows[off[i]:off[i+1]], dist[off[i]:off[i+1]]); !ok {
				return nil, false
			}
		}
		return results, true
	}
	k := args[0].K
	rows := make([]C.uint64_t, nq*k)
	dist := make([]C.double, nq*k)
	count := make([]C.int32_t, nq)
	if C.szg_search_topk(m.h, qp, C.int(nq), C.int(k), allow, &rows[0], &dist[0], &count[0]) != C.SZG_OK {
		return nil, false
	}
	for i := 0; i < nq; i++ {
		n := int(count[i])
		if results[i], ok = m.resultsOf(c, rows[i*k:i*k+n], dist[i*k:i*k+n]); !ok {
			return nil, false
		}
	}
	return results, true
}

// distancesTo is the gather-by-row primitive for candidate generators: the
// reference's float64 c.distance(query, doc.Vector) for each listed document, bit
// for bit (szg_distances).  The LSH path (lshtree.go:283-351) can collect a leaf's
// ids and score them in one call instead of one consider() per id.
func (m *gpuMirror) distancesTo(query []float64, ids []uint64) ([]float64, bool) {
	if m.dirty || len(ids) == 0 {
		return nil, false
	}
	rows := make([]C.uint64_t, len(ids))
	for i, id := range ids {
		row, ok := m.rowOf[id]
		if !ok {
			return nil, false
		}
		rows[i] = C.uint64_t(row)
	}
	out := make([]float64, len(ids))
	rc := C.szg_distances(m.h, (*C.double)(unsafe.Pointer(&query[0])), &rows[0], C.uint64_t(len(rows)),
		(*C.double)(unsafe.Pointer(&out[0])))
	return out, rc == C.SZG_OK
}

// pairDistances is c.distance(doc1.Vector, doc2.Vector) for stored documents
// (szg_pair_distances): computeAverageDistance (collection.go:348-400) keeps its
// rand.Intn pair selection and its in-order sum, and gets the distances from here.
func (m *gpuMirror) pairDistances(a, b []uint64) ([]float64, bool) {
	if m.dirty || len(a) == 0 || len(a) != len(b) {
		return nil, false
	}
	ra := make([]C.uint64_t, len(a))
	rb := make([]C.uint64_t, len(b))
	for i := range a {
		x, okA := m.rowOf[a[i]]
		y, okB := m.rowOf[b[i]]
		if !okA || !okB {
			return nil, false
		}
		ra[i], rb[i] = C.uint64_t(x), C.uint64_t(y)
	}
	out := make([]float64, len(a))
	rc := C.szg_pair_distances(m.h, &ra[0], &rb[0], C.uint64_t(len(a)), (*C.double)(unsafe.Pointer(&out[0])))
	return out, rc == C.SZG_OK
}


// searchIndexBulk is the default ("medium") path of Collection.Search -- c.index.search(args.Vector,
// radius, consider), collection.go:685-691 -- with the candidates' distances computed in bulk on
// the GPU.  The forest, its queue and consider() are the reference's, unchanged; only c.distance
// moves.  lshTree.search (lshtree.go:283-351) reads the candidates' distances back through two
// scalars only, `radius` (when a far-side leaf is popped, :305-310) and k_counter (:312-314);
// neither changes the ORDER in which nodes leave the queue, they only skip leaves and stop the
// walk.  So this function (1) pops the queue exactly as lshTree.search does but without pruning
// or stopping, buffering the next `window` candidates' leaves, (2) scores all their unvisited
// ids in ONE szg_distances call, and (3) replays the reference's loop body over the buffered
// leaves -- pruning test, stop test, visited marks, consider -- with those distances.  The
// result, its order and pointsSearched are the reference's on the same forest
// (tests/test_gpu_lsh.py drives the same algorithm, syzgydb_amd/lsh.py, against a C
// restatement of lshTree.search).  consider is Search's closure with its c.distance call
// replaced by the supplied value: considerWith(docid, distance, radius) (signal, radius).
func (m *gpuMirror) searchIndexBulk(c *Collection, tree *lshTree, vector []float64, radius float64,
	considerWith func(docid uint64, distance float64, radius float64) (int, float64)) bool {
	const window = 2048
	const search_k = 200
	if !m.ensureFresh(c, false) { // (row order is irrelevant here: candidates are addressed by id)
		return false
	}
	m.mu.RLock()
	defer m.mu.RUnlock()
	if m.dirty {
		return false // the caller runs the reference's own walk; nothing has been considered yet
	}
	considered := false // consider() has run: from here on a CPU re-walk would double-count
	length := vectorLength(vector)
	visited := make(map[uint64]bool)
	k_counter := 0
	pointAccepted := false
	pq := &nodePriorityQueue{}
	heap.Init(pq)
	for _, root := range tree.roots {
		heap.Push(pq, &nodePriorityItem{node: root, priority: 0})
	}
	type pending struct {
		node     *lshNode
		priority float64
	}
	for pq.Len() > 0 {
		// (1) the next leaves of the unpruned walk
		var leaves []pending
		var want []uint64
		queued := make(map[uint64]bool)
		for pq.Len() > 0 && len(want) < window {
			item := heap.Pop(pq).(*nodePriorityItem)
			node := item.node
			if !node.isLeaf() {
				dist, right := distanceToHyperplane(tree.c.DistanceMethod, vector, length, node.normal, node.b)
				if right {
					heap.Push(pq, &nodePriorityItem{node: node.right, priority: dist})
					heap.Push(pq, &nodePriorityItem{node: node.left, priority: -dist})
				} else {
					heap.Push(pq, &nodePriorityItem{node: node.left, priority: dist})
					heap.Push(pq, &nodePriorityItem{node: node.right, priority: -dist})
				}
				continue
			}
			leaves = append(leaves, pending{node, item.priority})
			for _, id := range node.ids {
				if !visited[id] && !queued[id] {
					queued[id] = true
					want = append(want, id)
				}
			}
		}
		// (2) one device call for the window
		distOf := make(map[uint64]float64, len(want))
		if len(want) > 0 {
			d, ok := m.distancesTo(vector, want)
			if !ok && !considered {
				return false // first window: the caller's closure state is untouched, it may walk on the CPU
			}
			for i, id := range want {
				if ok {
					distOf[id] = d[i]
					continue
				}
				// a later window failed on the device: finish THIS walk in place with the reference's own
				// c.distance (handing the walk back now would consider the earlier windows' documents twice)
				doc, err := c.getDocument(id)
				if err != nil {
					distOf[id] = math.NaN() // getDocument errors stop the reference's search (collection.go:585-587)
					continue
				}
				distOf[id] = c.distance(vector, doc.Vector)
			}
		}
		// (3) the reference's loop body over the buffered leaves
		for _, lf := range leaves {
			if lf.priority < 0 && -lf.priority > radius {
				continue // lshtree.go:305-310
			}
			if k_counter >= search_k {
				return true // :312-314
			}
			for _, id := range lf.node.ids {
				if visited[id] {
					continue
				}
				visited[id] = true
				considered = true
				var signal int
				signal, radius = considerWith(id, distOf[id], radius)
				switch signal {
				case StopSearch:
					return true
				case PointAccepted:
					k_counter = 0
					pointAccepted = true
				case PointChecked:
					if pointAccepted {
						k_counter++
					}
				case PointIgnored:
				}
			}
		}
	}
	return true
}

// ---- one process per GPU (SURVEY.md 8e) ------------------------------------------------------------------------
//
// A deployment that shards one collection's rows over the GPUs of a node runs one SyzgyDB process per GPU, each
// holding a contiguous range of the records (the visit order cut into ranges; szg_index_set_row_base makes rows
// global) and every query.  The exchange -- ONE ncclAllGather (RCCL over xGMI) of the per-rank top-(k+1) records
// per batch and the reference's selection replayed over the union -- happens inside the library; the host only
// hands the 128-byte communicator id from rank 0 to the others (its own RPC, a file, an environment variable).

type gpuComm struct{ h *C.szg_comm }

// commUniqueID is called on rank 0 only.
func commUniqueID() ([]byte, error) {
	id := make([]byte, C.SZG_COMM_ID_BYTES)
	if rc := C.szg_comm_unique_id((*C.uint8_t)(unsafe.Pointer(&id[0]))); rc != C.SZG_OK {
		return nil, fmt.Errorf("szg_comm_unique_id: %s", C.GoString(C.szg_last_error()))
	}
	return id, nil
}

// newGPUComm is collective: every rank calls it with the same id (ncclCommInitRank on `device`).
func newGPUComm(id []byte, rank, world, device int) (*gpuComm, error) {
	if len(id) != C.SZG_COMM_ID_BYTES {
		return nil, fmt.Errorf("communicator id must be %d bytes", C.SZG_COMM_ID_BYTES)
	}
	cm := &gpuComm{}
	rc := C.szg_comm_create(&cm.h, (*C.uint8_t)(unsafe.Pointer(&id[0])), C.int(rank), C.int(world), C.int(device))
	if rc != C.SZG_OK {
		return nil, fmt.Errorf("szg_comm_create: %s", C.GoString(C.szg_last_error()))
	}
	return cm, nil
}

func (cm *gpuComm) close() {
	if cm.h != nil {
		C.szg_comm_destroy(cm.h)
		cm.h = nil
	}
}

// attachComm: the mirror holds rows [rowBase, rowBase + len(ids)) of the sharded collection.
func (m *gpuMirror) attachComm(cm *gpuComm, rowBase uint64) bool {
	return C.szg_index_set_row_base(m.h, C.uint64_t(rowBase)) == C.SZG_OK &&
		C.szg_index_attach_comm(m.h, cm.h) == C.SZG_OK
}

// searchExactSharded is collective: every rank calls it with the same queries and K and gets the
// single-collection answer as GLOBAL rows (position in the unsharded visit order) with the reference's float64
// distances; the rank that owns a row resolves it to its document.  historyDependent[i] reports equal distances
// (or NaN) among the best K+1 of query i: the one case where the reference's order depends on its whole heap
// history, which no single rank holds -- the library then lets the heap travel rank 0 -> 1 -> ... in visit order,
// each rank replaying consider() over its own rows, so the answer is the unsharded collection's, order included.
func (m *gpuMirror) searchExactSharded(queries []float64, nq, k int) (rows []uint64, dist []float64, count []int32, historyDependent []bool, ok bool) {
	m.mu.RLock()
	defer m.mu.RUnlock()
	if m.dirty || nq <= 0 || k <= 0 || len(queries) == 0 || len(queries)%nq != 0 {
		return nil, nil, nil, nil, false
	}
	rows = make([]uint64, nq*k)
	dist = make([]float64, nq*k)
	count = make([]int32, nq)
	hist := make([]C.uint8_t, nq)
	rc := C.szg_search_topk_sharded(m.h, (*C.double)(unsafe.Pointer(&queries[0])), C.int(nq), C.int(k), nil,
		(*C.uint64_t)(unsafe.Pointer(&rows[0])), (*C.double)(unsafe.Pointer(&dist[0])),
		(*C.int32_t)(unsafe.Pointer(&count[0])), &hist[0])
	if rc != C.SZG_OK {
		return nil, nil, nil, nil, false
	}
	historyDependent = make([]bool, nq)
	for i := range hist {
		historyDependent[i] = hist[i] != 0
	}
	return rows, dist, count, historyDependent, true
}

// SearchBatch is the multi-query entry point SURVEY.md 8f-4 asks for beside the single-query Search
// (collection.go:569-711): the answers Search would give, query by query.  A list of exact searches of one kind --
// top-k with one K, or radius searches -- is ONE library call whose queries share sweeps of the corpus on the matrix
// cores (searchExactBatch); anything else is answered by Search itself, one by one.  INTEGRATION.md 1b shows the REST
// handler that exposes it beside handleSearch (rest.go:371-487).
func (c *Collection) SearchBatch(args []SearchArgs) []SearchResults {
	out := make([]SearchResults, len(args))
	batch := c.gpu != nil && len(args) > 1
	for i := range args {
		a := args[i]
		batch = batch && a.Precision == "exact" && (a.K > 0 || a.Radius > 0) &&
			(a.Radius > 0) == (args[0].Radius > 0) && (a.Radius > 0 || a.K == args[0].K)
	}
	if batch {
		c.mutex.RLock()
		res, ok := c.gpu.searchExactBatch(c, args, nil)
		_, numRecords := c.spanfile.GetStats()
		c.mutex.RUnlock()
		if ok {
			for i := range res {
				out[i] = SearchResults{Results: res[i], PercentSearched: 100}
				if numRecords == 0 { // collection.go:706-709
					out[i].PercentSearched = 0
				}
			}
			return out
		}
	}
	for i := range args {
		out[i] = c.Search(args[i])
	}
	return out
}
