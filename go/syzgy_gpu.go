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
	return nil
}

// add mirrors AddDocument (collection.go:427-457); called under c.mutex.Lock, so no search
// is in flight.  Rows are kept in IterateSortedRecords order (sort.Strings of the decimal ids,
// spanfile.go:540-560) because that order decides ties at the k boundary ("first visited
// wins", collection.go:608): a new id that sorts after every loaded one is appended in place;
// any other new id marks the mirror dirty and the next search reloads it in order.
func (m *gpuMirror) add(id uint64, encoded []byte) {
	m.version++
	p := (*C.uint8_t)(unsafe.Pointer(&encoded[0]))
	if row, ok := m.rowOf[id]; ok {
		if C.szg_index_overwrite(m.h, C.uint64_t(row), p) != C.SZG_OK {
			m.dirty = true
		}
		return
	}
	rid := strconv.FormatUint(id, 10)
	if m.dirty || rid < m.lastID {
		m.dirty = true // out of visit order: reload before the next exact search
		return
	}
	if C.szg_index_append(m.h, p, 1) != C.SZG_OK {
		m.dirty = true
		return
	}
	m.rowOf[id] = uint64(len(m.ids))
	m.ids = append(m.ids, id)
	m.lastID = rid
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

// searchExactKeyed: filterKey names args.Filter for the bitmask cache ("" = do not cache).
func (m *gpuMirror) searchExactKeyed(c *Collection, args SearchArgs, filterKey string) (results []SearchResult, ok bool) {
	m.mu.RLock()
	stale := m.dirty
	m.mu.RUnlock()
	if stale {
		// the caller holds only c.mutex.RLock: other searches may be inside the library right
		// now.  The mirror's own lock makes the reload exclusive; whoever gets it first reloads.
		m.mu.Lock()
		var err error
		if m.dirty {
			err = m.reload(c)
		}
		m.mu.Unlock()
		if err != nil {
			return nil, false
		}
	}
	m.mu.RLock()
	defer m.mu.RUnlock()
	if len(m.ids) == 0 {
		return make([]SearchResult, 0), true // empty collection: no results, [] not null (collection_test.go:294-309)
	}
	if len(args.Vector) != c.DimensionCount {
		return nil, false
	}
	var allow *C.uint64_t
	var keep []C.uint64_t
	if args.Filter != nil {
		keep = m.allowBitsKeyed(c, args.Filter, filterKey)
		allow = &keep[0]
	}
	q := (*C.double)(unsafe.Pointer(&args.Vector[0]))
	var rows []C.uint64_t
	var dist []C.double
	n := 0
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
				return nil, false
			}
			n = int(total)
			break
		}
	} else {
		rows = make([]C.uint64_t, args.K)
		dist = make([]C.double, args.K)
		var count C.int32_t
		rc := C.szg_search_topk(m.h, q, 1, C.int(args.K), allow, &rows[0], &dist[0], &count)
		if rc != C.SZG_OK {
			return nil, false
		}
		n = int(count)
	}
	results = make([]SearchResult, n)
	for i := 0; i < n; i++ {
		id := m.ids[rows[i]]
		span, err := c.spanfile.ReadRecord(fmt.Sprintf("%d", id))
		if err != nil {
			return nil, false
		}
		// Metadata is the live mmap slice, as in getDocument (collection.go:476); the
		// caller still holds c.mutex.RLock.
		results[i] = SearchResult{ID: id, Metadata: span.DataStreams[0].Data, Distance: float64(dist[i])}
	}
	_ = keep
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
	m.mu.RLock()
	defer m.mu.RUnlock()
	if m.dirty {
		return false // the caller runs the reference's own walk
	}
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
			if !ok {
				return false
			}
			for i, id := range want {
				distOf[id] = d[i]
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
