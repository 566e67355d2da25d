// executor_gpu.go — drop-in GPU executors for daviszhen/plan, binding libplanhip.so through cgo.
//
// Placement: pkg/compute/executor_gpu.go (package compute, so it sees the package's unexported
// identifiers: OperatorExec / OperatorResult (executor_operator.go:11-56), PhysicalOperator
// (builder_physical_operator.go:49-66), ScanOpInfo / JoinOpInfo / AggOpInfo (operator_info.go:10-34),
// Expr (expr.go:49-60), ensureOutputChunk (executor.go:201-210), newScanExecutor
// (executor_scan.go:52-60), NewExprExec / executeSelect / executeExprs (expr_exec.go:66-110, 342)).
// It references, besides those, only pkg/chunk, pkg/common, pkg/util, pkg/storage exported
// identifiers and the C symbols of include/planhip.h.
//
// NOT COMPILED HERE: this repository's build environment has no Go toolchain (SURVEY.md §8c). The
// file is a line-by-line Go rendering of plan_amd/csrc/host/executors.cpp, which IS compiled and
// tested (tests/test_host_layer.py: the reference's q1/q3/q6/q9 result files byte for byte through
// the same executor logic). The three `case` arms that select these executors are in
// INTEGRATION.md §2.
package compute

/*
#cgo CFLAGS: -I${SRCDIR}/../../third_party/planhip/include
#cgo LDFLAGS: -L${SRCDIR}/../../third_party/planhip/lib -lplanhip -Wl,-rpath,${SRCDIR}/../../third_party/planhip/lib
#include <stdlib.h>
#include <string.h>
#include "planhip.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"math/big"
	"os"
	"strconv"
	"strings"
	"sync"
	"time"
	"unsafe"

	decimal2 "github.com/govalues/decimal"

	"github.com/daviszhen/plan/pkg/chunk"
	"github.com/daviszhen/plan/pkg/common"
	"github.com/daviszhen/plan/pkg/storage"
	"github.com/daviszhen/plan/pkg/util"
)

// ---------------------------------------------------------------------------------- switches

// GPU use is opt-in per process: PLAN_GPU=1 [PLAN_GPU_DEVICE=n]. (A util.Config section would do the
// same; an environment switch keeps the change to this file and three case arms.)
var (
	gpuEnabled = os.Getenv("PLAN_GPU") == "1"
	gpuDevice  = func() int { d, _ := strconv.Atoi(os.Getenv("PLAN_GPU_DEVICE")); return d }()
)

// error model: a negative code + thread-local message becomes a Go error; nothing aborts.
// execQuery turns errors into a transaction rollback (executor_bench.go:184-204).
func phErr(rc C.int) error {
	if rc == C.PH_OK {
		return nil
	}
	return errors.New(C.GoString(C.ph_last_error()))
}

// errFallback: the sub-plan is outside the device path; the caller builds the CPU executor.
var errFallback = errors.New("planhip: shape outside the device path")

func fallbackIf(rc C.int) error {
	if rc == C.PH_EUNSUPPORTED || rc == C.PH_EOVERFLOW {
		return errFallback
	}
	return phErr(rc)
}

// one ph_ctx per query goroutine (calls on a ctx are stream-ordered, one thread at a time)
func newGpuCtx() (*C.ph_ctx, error) {
	var ctx *C.ph_ctx
	if err := phErr(C.ph_ctx_create(C.int(gpuDevice), &ctx)); err != nil {
		return nil, err
	}
	return ctx, nil
}

// ---------------------------------------------------------------------------------- lowering

// column position of a column reference inside its child chunk; which child: -1 -> children[0],
// -2 -> children[1], >= 0 -> this node (executeColumnRef, expr_exec.go:248-265)
func colRefOf(e *Expr) (table int64, col int, ok bool) {
	if e == nil || e.Typ != ET_Column {
		return 0, 0, false
	}
	return int64(e.ColRef.table()), int(e.ColRef.column()), true
}

func daysFromCivil(y, m, d int) int32 {
	return int32(time.Date(y, time.Month(m), d, 0, 0, 0, 0, time.UTC).Unix() / 86400)
}

func parseDateDays(s string) (int32, bool) {
	t, err := time.Parse("2006-01-02", strings.TrimSpace(s))
	if err != nil {
		return 0, false
	}
	return int32(t.Unix() / 86400), true
}

// decimal literal text -> (unscaled, scale)
func parseDecimalLiteral(s string) (int64, int, bool) {
	d, err := decimal2.Parse(strings.TrimSpace(s))
	if err != nil {
		return 0, 0, false
	}
	c := d.Coef()
	if c > (1<<63)-1 {
		return 0, 0, false
	}
	v := int64(c)
	if d.IsNeg() {
		v = -v
	}
	return v, d.Scale(), true
}

var cmpOps = map[string]C.int32_t{
	FuncEqual: C.PH_EQ, FuncNotEqual: C.PH_NE, FuncLess: C.PH_LT, FuncLessEqual: C.PH_LE,
	FuncGreater: C.PH_GT, FuncGreaterEqual: C.PH_GE, FuncLike: C.PH_LIKE, FuncNotLike: C.PH_NOTLIKE,
}

// strips casts the binder inserted around a column / literal (DecimalSizeCheck etc.)
func stripCast(e *Expr) *Expr {
	for e != nil && e.Typ == ET_Func && e.FuncName() == FuncCast && len(e.Children) >= 1 {
		e = e.Children[0]
	}
	return e
}

// literal -> ph_const, typed by the literal's bound type (the typing rules decide which compare
// kernel runs: a decimal-point literal is FLOAT = float32, builder_binder.go:264-273)
func lowerConst(e *Expr, keep *[]unsafe.Pointer) (C.ph_const, bool) {
	var k C.ph_const
	e = stripCast(e)
	if e == nil || e.Typ != ET_Const {
		return k, false
	}
	switch e.ConstValue.Type {
	case ConstTypeInteger:
		k._type = C.PH_I32
		k.i = C.int64_t(e.ConstValue.Integer)
	case ConstTypeFloat:
		k._type = C.PH_F32
		k.f = C.double(float32(e.ConstValue.Float)) // folded in float32 (rule_constant_folding.go:34-70)
	case ConstTypeDate:
		d, ok := parseDateDays(e.ConstValue.Date)
		if !ok {
			return k, false
		}
		k._type = C.PH_DATE
		k.i = C.int64_t(d)
	case ConstTypeDecimal:
		v, s, ok := parseDecimalLiteral(e.ConstValue.Decimal)
		if !ok {
			return k, false
		}
		k._type = C.PH_DEC64
		k.i = C.int64_t(v)
		k.scale = C.int32_t(s)
	case ConstTypeString:
		cs := C.CString(e.ConstValue.String)
		*keep = append(*keep, unsafe.Pointer(cs))
		k._type = C.PH_STR
		k.s = cs
	default:
		return k, false
	}
	return k, true
}

// one conjunct `column OP literal` (BETWEEN = two conjuncts) over child `wantTable`
func lowerConjunct(e *Expr, wantTable int64, keep *[]unsafe.Pointer) ([]C.ph_pred, bool) {
	if e == nil || e.Typ != ET_Func {
		return nil, false
	}
	name := e.FuncName()
	if name == FuncBetween && len(e.Children) == 3 {
		lo := &Expr{Typ: ET_Func, Info: e.Info, Children: []*Expr{e.Children[0], e.Children[1]}}
		_ = lo // BETWEEN is rewritten by the binder into >= AND <= before it reaches an executor
		return nil, false
	}
	if name == FuncAnd {
		var out []C.ph_pred
		for _, c := range e.Children {
			p, ok := lowerConjunct(c, wantTable, keep)
			if !ok {
				return nil, false
			}
			out = append(out, p...)
		}
		return out, true
	}
	op, isCmp := cmpOps[name]
	if !isCmp || len(e.Children) != 2 {
		return nil, false
	}
	tab, col, ok := colRefOf(stripCast(e.Children[0]))
	if !ok || tab != wantTable {
		return nil, false
	}
	k, ok := lowerConst(e.Children[1], keep)
	if !ok {
		return nil, false
	}
	var p C.ph_pred
	p.col = C.int32_t(col)
	p.op = op
	p.k = k
	return []C.ph_pred{p}, true
}

// argument expression -> RPN over the columns of child `wantTable` (ph_rpn); the decimal typing
// (Mul: scales add, Add/Sub: max) is the library's (ph_expr_scale)
func lowerRPN(e *Expr, wantTable int64, out *[]C.ph_rpn) bool {
	e = stripCast(e)
	if e == nil {
		return false
	}
	switch e.Typ {
	case ET_Column:
		tab, col, _ := colRefOf(e)
		if tab != wantTable {
			return false
		}
		*out = append(*out, C.ph_rpn{op: C.PH_X_COL, col: C.int32_t(col)})
		return true
	case ET_Const:
		switch e.ConstValue.Type {
		case ConstTypeInteger:
			*out = append(*out, C.ph_rpn{op: C.PH_X_CONST, col: -1, ival: C.int64_t(e.ConstValue.Integer)})
			return true
		case ConstTypeDecimal:
			v, s, ok := parseDecimalLiteral(e.ConstValue.Decimal)
			if !ok {
				return false
			}
			*out = append(*out, C.ph_rpn{op: C.PH_X_CONST, col: -1, ival: C.int64_t(v), scale: C.int32_t(s)})
			return true
		}
		return false
	case ET_Func:
		var op C.int32_t
		switch e.FuncName() {
		case FuncAdd:
			op = C.PH_X_ADD
		case FuncSubtract:
			op = C.PH_X_SUB
		case FuncMultiply:
			op = C.PH_X_MUL
		default:
			return false
		}
		if len(e.Children) != 2 || !lowerRPN(e.Children[0], wantTable, out) || !lowerRPN(e.Children[1], wantTable, out) {
			return false
		}
		*out = append(*out, C.ph_rpn{op: op, col: -1})
		return true
	}
	return false
}

var aggKinds = map[string]C.int32_t{"sum": C.PH_A_SUM, "avg": C.PH_A_AVG, "count": C.PH_A_COUNT, "min": C.PH_A_MIN, "max": C.PH_A_MAX}

// AggOpInfo.Aggs -> ph_aggexpr list (argument as RPN, count(*) without one)
func lowerAggs(aggs []*Expr, wantTable int64) ([]C.ph_aggexpr, bool) {
	out := make([]C.ph_aggexpr, 0, len(aggs))
	for _, a := range aggs {
		if a == nil || a.Typ != ET_Func || a.GetFuncInfo().FunImpl == nil || a.GetFuncInfo().FunImpl.IsDistinct() {
			return nil, false // DISTINCT aggregates: see INTEGRATION.md §3 (ph_agg_sink_masked); CPU here
		}
		kind, ok := aggKinds[strings.ToLower(a.FuncName())]
		if !ok {
			return nil, false
		}
		var ax C.ph_aggexpr
		if kind == C.PH_A_COUNT && (len(a.Children) == 0 || stripCast(a.Children[0]).Typ == ET_Const) {
			ax.kind = C.PH_A_COUNT_STAR
		} else {
			var prog []C.ph_rpn
			if len(a.Children) != 1 || !lowerRPN(a.Children[0], wantTable, &prog) || len(prog) > 12 {
				return nil, false
			}
			ax.kind = kind
			ax.nprog = C.int32_t(len(prog))
			for i := range prog {
				ax.prog[i] = prog[i]
			}
		}
		out = append(out, ax)
	}
	return out, true
}

// ---------------------------------------------------------------------------------- staging

// cBuf: a growing C.malloc'd buffer. Vector.Data is Go heap (`make([]byte)`, vec_buffer.go:30-35), so
// it may only be passed to C for the duration of a call and cannot be pinned: batches are copied
// once into C memory, which the library stages through its own pinned buffers.
type cBuf struct {
	p   unsafe.Pointer
	len int
	cap int
}

func (b *cBuf) grow(n int) unsafe.Pointer {
	if b.len+n > b.cap {
		nc := b.cap*2 + n + 4096
		b.p = C.realloc(b.p, C.size_t(nc))
		b.cap = nc
	}
	at := unsafe.Add(b.p, b.len)
	b.len += n
	return at
}
func (b *cBuf) reset() { b.len = 0 }
func (b *cBuf) free()  { C.free(b.p); b.p = nil; b.cap = 0; b.len = 0 }

// stagedCol: one column of many chunks in the device encoding (SURVEY.md §8d): INTEGER int32,
// BIGINT int64, DATE int32 days, DECIMAL int64 unscaled at the type's scale, HUGEINT int64 (scale-0
// decimal: its only comparison is '>'), VARCHAR uint8 dictionary code (<= 256 values) or, when
// asString, int32 offsets + bytes.
type stagedCol struct {
	typ       common.LType
	asString  bool
	data      cBuf
	bytes     cBuf
	valid     []uint8
	hasNull   bool
	dict      []string
	dictIndex map[string]int
	// value range of the non-NULL integer values staged so far: the column statistics ph_join_build_ex
	// wants for a direct (dense-key) join table. Staging touches every value anyway.
	lo, hi  int64
	ranged  bool
}

func (sc *stagedCol) note(v int64) {
	if !sc.ranged {
		sc.lo, sc.hi, sc.ranged = v, v, true
	} else if v < sc.lo {
		sc.lo = v
	} else if v > sc.hi {
		sc.hi = v
	}
}

func phTypeOf(t common.LType, asString bool) (C.int32_t, int, bool) {
	switch t.GetInternalType() {
	case common.INT32:
		return C.PH_I32, 4, true
	case common.INT64:
		return C.PH_I64, 8, true
	case common.DATE:
		return C.PH_DATE, 4, true
	case common.DECIMAL, common.INT128:
		return C.PH_DEC64, 8, true
	case common.VARCHAR:
		if asString {
			return C.PH_STR, 4, true
		}
		return C.PH_CODE8, 1, true
	}
	return 0, 0, false
}

func decimalUnscaled(d *common.Decimal, scale int) (int64, bool) {
	c := d.Coef()
	for s := d.Scale(); s < scale; s++ {
		if c > (1<<63-1)/10 {
			return 0, false
		}
		c *= 10
	}
	if d.Scale() > scale || c > 1<<63-1 {
		return 0, false
	}
	if d.IsNeg() {
		return -int64(c), true
	}
	return int64(c), true
}

func (sc *stagedCol) append(vec *chunk.Vector, card int, rowBase int) error {
	var uni chunk.UnifiedFormat
	vec.ToUnifiedFormat(card, &uni) // FLAT / CONST / DICT / SEQUENCE -> data + sel + mask (vector_format.go:64-97)
	need := (rowBase + card + 7) / 8
	for len(sc.valid) < need {
		sc.valid = append(sc.valid, 0)
	}
	_, w, _ := phTypeOf(sc.typ, sc.asString)
	for i := 0; i < card; i++ {
		idx := uni.Sel.GetIndex(i)
		row := rowBase + i
		ok := uni.Mask.RowIsValid(uint64(idx))
		if ok {
			sc.valid[row>>3] |= 1 << uint(row&7)
		} else {
			sc.hasNull = true
		}
		if sc.asString {
			if sc.data.len == 0 {
				*(*int32)(sc.data.grow(4)) = 0
			}
			if ok {
				s := chunk.GetSliceInPhyFormatUnifiedFormat[common.String](&uni)[idx]
				C.memcpy(sc.bytes.grow(s.Len), s.Data, C.size_t(s.Len))
			}
			*(*int32)(sc.data.grow(4)) = int32(sc.bytes.len)
			continue
		}
		dst := sc.data.grow(w)
		if !ok {
			C.memset(dst, 0, C.size_t(w))
			continue
		}
		switch sc.typ.GetInternalType() {
		case common.INT32:
			v := chunk.GetSliceInPhyFormatUnifiedFormat[int32](&uni)[idx]
			*(*int32)(dst) = v
			sc.note(int64(v))
		case common.INT64:
			v := chunk.GetSliceInPhyFormatUnifiedFormat[int64](&uni)[idx]
			*(*int64)(dst) = v
			sc.note(v)
		case common.DATE:
			d := chunk.GetSliceInPhyFormatUnifiedFormat[common.Date](&uni)[idx]
			*(*int32)(dst) = daysFromCivil(int(d.Year), int(d.Month), int(d.Day))
		case common.DECIMAL:
			d := chunk.GetSliceInPhyFormatUnifiedFormat[common.Decimal](&uni)[idx]
			v, fits := decimalUnscaled(&d, sc.typ.Scale)
			if !fits {
				return errFallback
			}
			*(*int64)(dst) = v
		case common.INT128:
			h := chunk.GetSliceInPhyFormatUnifiedFormat[common.Hugeint](&uni)[idx]
			if h.Upper != int64(h.Lower)>>63 {
				return errFallback
			}
			*(*int64)(dst) = int64(h.Lower)
		case common.VARCHAR:
			s := chunk.GetSliceInPhyFormatUnifiedFormat[common.String](&uni)[idx]
			key := s.String()
			code, has := sc.dictIndex[key]
			if !has {
				if len(sc.dict) >= 256 {
					return errFallback
				}
				code = len(sc.dict)
				sc.dict = append(sc.dict, key)
				sc.dictIndex[key] = code
			}
			*(*uint8)(dst) = uint8(code)
		}
	}
	return nil
}

// deviceBatch: selected columns of many child chunks, uploaded as device columns
type deviceBatch struct {
	ctx   *C.ph_ctx
	cols  []int
	sc    []*stagedCol
	rows  int
	dev   []C.ph_col
	owned []unsafe.Pointer
}

func newDeviceBatch(ctx *C.ph_ctx, types []common.LType, cols []int, asString []bool) (*deviceBatch, error) {
	b := &deviceBatch{ctx: ctx, cols: cols}
	for k, c := range cols {
		if _, _, ok := phTypeOf(types[c], len(asString) > k && asString[k]); !ok {
			return nil, errFallback
		}
		b.sc = append(b.sc, &stagedCol{typ: types[c], asString: len(asString) > k && asString[k], dictIndex: map[string]int{}})
	}
	return b, nil
}

func (b *deviceBatch) append(c *chunk.Chunk) error {
	for k, col := range b.cols {
		if err := b.sc[k].append(c.Data[col], c.Card(), b.rows); err != nil {
			return err
		}
	}
	b.rows += c.Card()
	return nil
}

func (b *deviceBatch) releaseDevice() {
	for _, p := range b.owned {
		C.ph_dev_free(b.ctx, p)
	}
	b.owned, b.dev = nil, nil
}

func (b *deviceBatch) reset() {
	b.releaseDevice()
	for _, sc := range b.sc {
		sc.data.reset()
		sc.bytes.reset()
		sc.valid = sc.valid[:0]
		sc.hasNull = false
		sc.ranged = false
	}
	b.rows = 0
}

func (b *deviceBatch) close() {
	b.releaseDevice()
	for _, sc := range b.sc {
		sc.data.free()
		sc.bytes.free()
	}
}

func (b *deviceBatch) devAlloc(host unsafe.Pointer, n int) (unsafe.Pointer, error) {
	var d unsafe.Pointer
	if err := phErr(C.ph_dev_alloc(b.ctx, C.int64_t(n+64), &d)); err != nil {
		return nil, err
	}
	b.owned = append(b.owned, d)
	if n > 0 {
		if err := phErr(C.ph_dev_upload(b.ctx, d, host, C.int64_t(n))); err != nil {
			return nil, err
		}
	}
	return d, nil
}

func (b *deviceBatch) upload() error {
	b.releaseDevice()
	b.dev = make([]C.ph_col, len(b.sc))
	for k, sc := range b.sc {
		t, _, _ := phTypeOf(sc.typ, sc.asString)
		d, err := b.devAlloc(sc.data.p, sc.data.len)
		if err != nil {
			return err
		}
		col := C.ph_col{_type: t, data: d}
		if sc.typ.GetInternalType() != common.INT128 {
			col.scale = C.int32_t(sc.typ.Scale)
		}
		if sc.asString {
			a, err := b.devAlloc(sc.bytes.p, sc.bytes.len)
			if err != nil {
				return err
			}
			col.aux = a
			col.aux_bytes = C.int64_t(sc.bytes.len)
		}
		if sc.hasNull {
			v, err := b.devAlloc(unsafe.Pointer(&sc.valid[0]), (b.rows+7)/8)
			if err != nil {
				return err
			}
			col.validity = (*C.uint8_t)(v)
		}
		b.dev[k] = col
	}
	return nil
}

func (b *deviceBatch) codeOf(k int, s string) int {
	if c, ok := b.sc[k].dictIndex[s]; ok {
		return c
	}
	return 999 // matches nothing
}

// ---------------------------------------------------------------------------------- resident tables

// residentTable: the pruned columns of a scanned table, loaded once into HBM (ph_table_create) and
// shared read-only by every query — what replaces DataTable.Scan's per-chunk materialisation
// (pkg/storage/table.go:418-428 feeding scanRows, executor_scan.go:158-241).
type residentTable struct {
	h     *C.ph_table
	types []common.LType
	dicts [][]string
}

var (
	residentMu     sync.Mutex
	residentTables = map[string]*residentTable{}
	residentCtx    *C.ph_ctx // tables outlive queries: they live on a process-wide ctx
)

func residentTableFor(scanOp *PhysicalOperator, cfg *util.Config, txn *storage.Txn) (*residentTable, error) {
	key := scanOp.getScanDatabase() + "." + scanOp.getScanTable() + "#" + strings.Join(scanOp.getScanColumns(), ",")
	residentMu.Lock()
	defer residentMu.Unlock()
	if t, ok := residentTables[key]; ok {
		return t, nil
	}
	if residentCtx == nil {
		var err error
		if residentCtx, err = newGpuCtx(); err != nil {
			return nil, err
		}
	}
	// pull the table once through the reference's own scan, WITHOUT its pushed-down filter
	// (the filter runs on the device per query)
	bare := *scanOp
	bare.Filters = nil
	scan, err := newScanExecutor(&bare, cfg, txn, nil)
	if err != nil {
		return nil, err
	}
	if err = scan.Init(); err != nil {
		return nil, err
	}
	defer scan.Close()
	types := make([]common.LType, len(bare.Outputs))
	cols := make([]int, len(bare.Outputs))
	for i, o := range bare.Outputs {
		types[i] = o.DataTyp
		cols[i] = i
	}
	batch, err := newDeviceBatch(residentCtx, types, cols, nil)
	if err != nil {
		return nil, err
	}
	defer batch.close()
	for {
		c := &chunk.Chunk{}
		res, err := scan.Execute(nil, c)
		if err != nil {
			return nil, err
		}
		if res == Done {
			break
		}
		if c.Card() == 0 {
			continue
		}
		if err = batch.append(c); err != nil {
			return nil, err
		}
	}
	host := make([]C.ph_col, len(cols))
	var keep []unsafe.Pointer
	defer func() {
		for _, p := range keep {
			C.free(p)
		}
	}()
	rt := &residentTable{types: types, dicts: make([][]string, len(cols))}
	for k, sc := range batch.sc {
		t, _, _ := phTypeOf(sc.typ, false)
		host[k] = C.ph_col{_type: t, scale: C.int32_t(sc.typ.Scale), data: sc.data.p}
		if sc.hasNull {
			host[k].validity = (*C.uint8_t)(unsafe.Pointer(&sc.valid[0]))
		}
		if t == C.PH_CODE8 { // dictionary: NUL-separated strings
			blob := strings.Join(sc.dict, "\x00") + "\x00"
			p := C.CBytes([]byte(blob))
			keep = append(keep, p)
			host[k].aux = p
			host[k].aux_bytes = C.int64_t(len(blob))
			rt.dicts[k] = append([]string(nil), sc.dict...)
		}
	}
	if err = fallbackIf(C.ph_table_create(residentCtx, C.int32_t(len(host)), &host[0], C.int64_t(batch.rows), &rt.h)); err != nil {
		return nil, err
	}
	// The catalog's PRIMARY KEY / UNIQUE constraints become declared-unique column sets of the resident table: what
	// lets a resident plan (executor_gpu_plan.go) run a join against this table as an N:1 lookup. Constraint fields are
	// unexported in pkg/storage; INTEGRATION.md §2 adds the 10-line accessor CatalogEntry.UniqueKeys() [][]string.
	// Order statistics (ascending / strictly ascending per column) the library gathers itself at ph_table_create.
	if si, ok := scanOp.Info.(*ScanOpInfo); ok && si.TableEnt != nil {
		pos := map[string]int{}
		for i, name := range scanOp.getScanColumns() {
			pos[name] = i
		}
		for _, uk := range si.TableEnt.UniqueKeys() {
			cols := make([]C.int32_t, 0, len(uk))
			for _, name := range uk {
				if p, ok := pos[name]; ok {
					cols = append(cols, C.int32_t(p))
				}
			}
			if len(cols) == len(uk) && len(cols) > 0 { // only when every key column was loaded
				if err = phErr(C.ph_table_declare_unique(rt.h, C.int32_t(len(cols)), &cols[0])); err != nil {
					return nil, err
				}
			}
		}
	}
	residentTables[key] = rt
	return rt, nil
}

// ---------------------------------------------------------------------------------- finalize

func decFromInt128(lo uint64, hi int64, scale int) (common.Decimal, error) {
	v := new(big.Int).SetInt64(hi)
	v.Lsh(v, 64).Add(v, new(big.Int).SetUint64(lo))
	if v.IsInt64() {
		d, err := decimal2.New(v.Int64(), scale)
		return common.Decimal{Decimal: d}, err
	}
	pow := new(big.Int).Exp(big.NewInt(10), big.NewInt(int64(scale)), nil)
	whole, frac := new(big.Int).QuoRem(v, pow, new(big.Int))
	if !whole.IsInt64() {
		return common.Decimal{}, errors.New("decimal sum exceeds 19 digits")
	}
	d, err := decimal2.NewFromInt64(whole.Int64(), frac.Int64(), scale)
	return common.Decimal{Decimal: d}, err
}

// emitGroupRows writes groups [from, from+n) of a device result into `out` with the reference's
// FinalizeStates typing (function_aggr.go:1330-1365): SUM -> HUGEINT / DECIMAL or NULL when never
// set (SumOp.Finalize :813-823), AVG -> sum.Quo(count) / float64 division (AvgOp.Finalize :873-900),
// COUNT -> HUGEINT or NULL when 0 (CountOp.Finalize :950-962), MIN/MAX -> the argument's type.
func emitGroupRows(out *chunk.Chunk, keyTypes []common.LType, keyDicts [][]string, kinds []C.int32_t,
	argTypes []common.LType, argScales []int, nk, na int, keys []int64, keyNull []uint8,
	lo []uint64, hi []int64, cnt []uint64, from, n int) error {
	for r := 0; r < n; r++ {
		g := from + r
		for c := 0; c < len(keyTypes); c++ {
			v := out.Data[c]
			if keyNull != nil && keyNull[g*nk+c] != 0 {
				chunk.SetNullInPhyFormatFlat(v, uint64(r), true)
				continue
			}
			kv := keys[g*nk+c]
			switch keyTypes[c].GetInternalType() {
			case common.INT32:
				chunk.GetSliceInPhyFormatFlat[int32](v)[r] = int32(kv)
			case common.INT64:
				chunk.GetSliceInPhyFormatFlat[int64](v)[r] = kv
			case common.DATE:
				t := time.Unix(kv*86400, 0).UTC()
				chunk.GetSliceInPhyFormatFlat[common.Date](v)[r] = common.Date{Year: int32(t.Year()), Month: int32(t.Month()), Day: int32(t.Day())}
			case common.DECIMAL:
				d, err := decimal2.New(kv, keyTypes[c].Scale)
				if err != nil {
					return err
				}
				chunk.GetSliceInPhyFormatFlat[common.Decimal](v)[r] = common.Decimal{Decimal: d}
			case common.VARCHAR:
				s := keyDicts[c][kv]
				p := util.CMalloc(len(s)) // VARCHAR cells hold C-malloc'd bytes (vector.go:208-217)
				util.PointerCopy(p, unsafe.Pointer(unsafe.StringData(s)), len(s))
				chunk.GetSliceInPhyFormatFlat[common.String](v)[r] = common.String{Len: len(s), Data: p}
			}
		}
		for a := 0; a < na; a++ {
			v := out.Data[len(keyTypes)+a]
			si := g*na + a
			cn := cnt[si]
			if cn == 0 {
				chunk.SetNullInPhyFormatFlat(v, uint64(r), true)
				continue
			}
			dec := argTypes[a].Id == common.LTID_DECIMAL
			switch kinds[a] {
			case C.PH_A_SUM:
				if dec {
					d, err := decFromInt128(lo[si], hi[si], argScales[a])
					if err != nil {
						return err
					}
					chunk.GetSliceInPhyFormatFlat[common.Decimal](v)[r] = d
				} else {
					chunk.GetSliceInPhyFormatFlat[common.Hugeint](v)[r] = common.Hugeint{Lower: lo[si], Upper: hi[si]}
				}
			case C.PH_A_AVG:
				if dec {
					s, err := decFromInt128(lo[si], hi[si], argScales[a])
					if err != nil {
						return err
					}
					q, err := s.Decimal.Quo(decimal2.MustNew(int64(cn), 0))
					if err != nil {
						return err
					}
					chunk.GetSliceInPhyFormatFlat[common.Decimal](v)[r] = common.Decimal{Decimal: q}
				} else {
					f := new(big.Float).SetInt(new(big.Int).Add(new(big.Int).Lsh(big.NewInt(hi[si]), 64), new(big.Int).SetUint64(lo[si])))
					sum, _ := f.Float64()
					chunk.GetSliceInPhyFormatFlat[float64](v)[r] = sum / float64(cn)
				}
			case C.PH_A_COUNT, C.PH_A_COUNT_STAR, C.PH_A_COUNT_DISTINCT:
				chunk.GetSliceInPhyFormatFlat[common.Hugeint](v)[r] = common.Hugeint{Lower: cn}
			case C.PH_A_MIN, C.PH_A_MAX:
				if dec {
					d, err := decimal2.New(int64(lo[si]), argScales[a])
					if err != nil {
						return err
					}
					chunk.GetSliceInPhyFormatFlat[common.Decimal](v)[r] = common.Decimal{Decimal: d}
				} else if argTypes[a].GetInternalType() == common.INT32 {
					chunk.GetSliceInPhyFormatFlat[int32](v)[r] = int32(int64(lo[si]))
				} else {
					chunk.GetSliceInPhyFormatFlat[int64](v)[r] = int64(lo[si])
				}
			}
		}
	}
	out.SetCard(n)
	return nil
}

// ---------------------------------------------------------------------------------- Agg <- Scan(filter)

// gpuScanAggExecutor collapses  Agg <- Scan(filter)  into one ph_scan_plan over the resident table
// (fused kernel: the precompiled Q1 / Q6 shapes, a kernel generated for the plan, or the operator
// chain). HAVING and the output expressions run through the reference's own ExprExec over the
// finalised group rows, exactly as aggExecutor's output phase does (executor_aggr.go:143-263).
type gpuScanAggExecutor struct {
	op       *PhysicalOperator
	cfg      *util.Config
	txn      *storage.Txn
	ctx      *C.ph_ctx
	table    *residentTable
	plan     *C.ph_scan_plan
	result   *C.ph_agg_result
	next     int
	groupIdx []int
	kinds    []C.int32_t
	argTypes []common.LType
	rowTypes []common.LType // [group columns | aggregate results]
	havingEx *ExprExec
	outputEx *ExprExec
	sel      *chunk.SelectVector
}

func tryNewGpuScanAggExecutor(op *PhysicalOperator, cfg *util.Config, txn *storage.Txn, children []OperatorExec) (OperatorExec, error) {
	if !gpuEnabled || len(op.Children) != 1 || op.Children[0].Typ != POT_Scan || op.Children[0].getScanTyp() != ScanTypeTable {
		return nil, errFallback
	}
	info, ok := op.Info.(*AggOpInfo)
	if !ok {
		return nil, errFallback
	}
	e := &gpuScanAggExecutor{op: op, cfg: cfg, txn: txn}
	var keep []unsafe.Pointer
	defer func() {
		for _, p := range keep {
			C.free(p)
		}
	}()
	scanOp := op.Children[0]
	var preds []C.ph_pred
	for _, f := range scanOp.Filters { // pushed-down conjuncts over the scan's own columns
		p, ok := lowerConjunct(f, int64(scanOp.Index), &keep)
		if !ok {
			p, ok = lowerConjunct(f, -1, &keep)
		}
		if !ok {
			return nil, errFallback
		}
		preds = append(preds, p...)
	}
	var groups []C.int32_t
	for _, g := range info.GroupBys {
		tab, col, ok := colRefOf(stripCast(g))
		if !ok || tab != -1 {
			if stripCast(g).Typ == ET_Const { // constant key of an ungrouped aggregate (executor_aggr.go:37-48)
				continue
			}
			return nil, errFallback
		}
		groups = append(groups, C.int32_t(col))
		e.groupIdx = append(e.groupIdx, col)
	}
	aggs, ok := lowerAggs(info.Aggs, -1)
	if !ok {
		return nil, errFallback
	}
	var err error
	if e.table, err = residentTableFor(scanOp, cfg, txn); err != nil {
		return nil, err
	}
	if e.ctx, err = newGpuCtx(); err != nil {
		return nil, err
	}
	pp, gp := (*C.ph_pred)(nil), (*C.int32_t)(nil)
	if len(preds) > 0 {
		pp = &preds[0]
	}
	if len(groups) > 0 {
		gp = &groups[0]
	}
	if err = fallbackIf(C.ph_scan_plan_create(e.ctx, e.table.h, pp, C.int32_t(len(preds)), gp, C.int32_t(len(groups)),
		&aggs[0], C.int32_t(len(aggs)), &e.plan)); err != nil {
		C.ph_ctx_destroy(e.ctx)
		return nil, err
	}
	for _, g := range e.groupIdx {
		e.rowTypes = append(e.rowTypes, e.table.types[g])
	}
	for i, a := range info.Aggs {
		e.kinds = append(e.kinds, aggs[i].kind)
		at := common.IntegerType()
		if aggs[i].kind != C.PH_A_COUNT_STAR {
			at = a.Children[0].DataTyp
		}
		e.argTypes = append(e.argTypes, at)
		e.rowTypes = append(e.rowTypes, a.DataTyp)
	}
	_ = children // the CPU scan executor built for the child is not pulled; buildOperatorExec closes it with the tree
	return e, nil
}

func (e *gpuScanAggExecutor) Init() error {
	if len(e.op.Filters) > 0 {
		e.havingEx = NewExprExec(e.op.Filters...)
		e.sel = chunk.NewSelectVector(util.DefaultVectorSize)
	}
	e.outputEx = NewExprExec(e.op.Outputs...)
	return nil
}

func (e *gpuScanAggExecutor) Execute(input, output *chunk.Chunk) (OperatorResult, error) {
	ensureOutputChunk(e.op, output)
	if e.result == nil { // pipeline breaker, like HAS_INIT (executor_aggr.go:110-142)
		if err := phErr(C.ph_scan_plan_run(e.plan, 0, C.ph_table_rows(e.table.h))); err != nil {
			return InvalidOpResult, err
		}
		if err := phErr(C.ph_scan_plan_fetch(e.plan, &e.result)); err != nil {
			return InvalidOpResult, err
		}
	}
	r := e.result
	ng, nk, na := int(r.ngroups), int(r.nkeys), int(r.naggs)
	for e.next < ng {
		n := ng - e.next
		if n > util.DefaultVectorSize {
			n = util.DefaultVectorSize
		}
		rows := &chunk.Chunk{}
		rows.Init(e.rowTypes, util.DefaultVectorSize)
		keys := unsafe.Slice((*int64)(unsafe.Pointer(r.keys)), ng*maxInt(nk, 1))
		lo := unsafe.Slice((*uint64)(unsafe.Pointer(r.sum_lo)), ng*na)
		hi := unsafe.Slice((*int64)(unsafe.Pointer(r.sum_hi)), ng*na)
		cnt := unsafe.Slice((*uint64)(unsafe.Pointer(r.count)), ng*na)
		scales := make([]int, na)
		for a := 0; a < na; a++ {
			scales[a] = int(*(*C.int32_t)(unsafe.Add(unsafe.Pointer(r.scale), 4*a)))
		}
		keyTypes := e.rowTypes[:len(e.groupIdx)]
		dicts := make([][]string, len(e.groupIdx))
		for i, g := range e.groupIdx {
			dicts[i] = e.table.dicts[g]
		}
		if err := emitGroupRows(rows, keyTypes, dicts, e.kinds, e.argTypes, scales, maxInt(nk, 1), na, keys, nil, lo, hi, cnt, e.next, n); err != nil {
			return InvalidOpResult, err
		}
		e.next += n
		// output phase: HAVING over the aggregate results, then the output expressions
		// ([]*Chunk{child columns = group columns, nil, aggregate results}, executor_aggr.go:176-247)
		groupChunk, aggrChunk := &chunk.Chunk{}, &chunk.Chunk{}
		groupChunk.Init(e.rowTypes[:len(e.groupIdx)], util.DefaultVectorSize)
		aggrChunk.Init(e.rowTypes[len(e.groupIdx):], util.DefaultVectorSize)
		for i := range e.groupIdx {
			groupChunk.Data[i].Reference(rows.Data[i])
		}
		for i := 0; i < na; i++ {
			aggrChunk.Data[i].Reference(rows.Data[len(e.groupIdx)+i])
		}
		groupChunk.SetCard(n)
		aggrChunk.SetCard(n)
		if e.havingEx != nil {
			count, err := e.havingEx.executeSelect([]*chunk.Chunk{groupChunk, nil, aggrChunk}, e.sel)
			if err != nil {
				return InvalidOpResult, err
			}
			if count == 0 {
				continue
			}
			if count != n {
				groupChunk.SliceItself(e.sel, count)
				aggrChunk.SliceItself(e.sel, count)
			}
		}
		if err := e.outputEx.executeExprs([]*chunk.Chunk{groupChunk, nil, aggrChunk}, output); err != nil {
			return InvalidOpResult, err
		}
		if output.Card() > 0 {
			return haveMoreOutput, nil
		}
	}
	return Done, nil
}

func (e *gpuScanAggExecutor) Close() error {
	if e.result != nil {
		C.ph_agg_result_free(e.result)
	}
	if e.plan != nil {
		C.ph_scan_plan_free(e.plan)
	}
	if e.ctx != nil {
		C.ph_ctx_destroy(e.ctx)
	}
	e.result, e.plan, e.ctx = nil, nil, nil
	return nil
}

func maxInt(a, b int) int {
	if a > b {
		return a
	}
	return b
}

// ---------------------------------------------------------------------------------- Filter

// gpuFilterExecutor <- filterExecutor (executor_filter.go:27-114): batches child chunks, runs one
// ph_filter_select per conjunct (each narrowing the previous selection: execSelectAnd,
// expr_exec.go:444-486), splits the batch selection back into per-chunk SelectVectors and emits
// SliceIndice views (no row is copied).
type gpuFilterExecutor struct {
	op        *PhysicalOperator
	child     OperatorExec
	ctx       *C.ph_ctx
	preds     []C.ph_pred
	keep      []unsafe.Pointer
	cols      []int
	batch     *deviceBatch
	ready     []readyChunk
	childDone bool
}

type readyChunk struct {
	c   *chunk.Chunk
	sel *chunk.SelectVector
	n   int
}

const gpuBatchChunks = 512

func tryNewGpuFilterExecutor(op *PhysicalOperator, cfg *util.Config, txn *storage.Txn, children []OperatorExec) (OperatorExec, error) {
	if !gpuEnabled || len(children) != 1 {
		return nil, errFallback
	}
	e := &gpuFilterExecutor{op: op, child: children[0]}
	for _, f := range op.Filters {
		p, ok := lowerConjunct(f, -1, &e.keep)
		if !ok {
			e.freeKeep()
			return nil, errFallback
		}
		e.preds = append(e.preds, p...)
	}
	var asString []bool
	for _, p := range e.preds {
		k := -1
		for i, c := range e.cols {
			if c == int(p.col) {
				k = i
			}
		}
		if k < 0 {
			e.cols = append(e.cols, int(p.col))
			asString = append(asString, false)
			k = len(e.cols) - 1
		}
		if p.op == C.PH_LIKE || p.op == C.PH_NOTLIKE {
			asString[k] = true
		}
	}
	types := make([]common.LType, len(op.Children[0].Outputs))
	for i, o := range op.Children[0].Outputs {
		types[i] = o.DataTyp
	}
	var err error
	if e.ctx, err = newGpuCtx(); err != nil {
		e.freeKeep()
		return nil, err
	}
	if e.batch, err = newDeviceBatch(e.ctx, types, e.cols, asString); err != nil {
		C.ph_ctx_destroy(e.ctx)
		e.freeKeep()
		return nil, err
	}
	return e, nil
}

func (e *gpuFilterExecutor) freeKeep() {
	for _, p := range e.keep {
		C.free(p)
	}
	e.keep = nil
}

func (e *gpuFilterExecutor) Init() error { return nil }

func (e *gpuFilterExecutor) fill() error {
	var chunks []*chunk.Chunk
	e.batch.reset()
	for len(chunks) < gpuBatchChunks {
		c := &chunk.Chunk{}
		res, err := e.child.Execute(nil, c)
		if err != nil {
			return err
		}
		if res == InvalidOpResult {
			return errors.New("child failed")
		}
		if res == Done {
			e.childDone = true
			break
		}
		if c.Card() == 0 {
			continue
		}
		if err = e.batch.append(c); err != nil {
			return err
		}
		chunks = append(chunks, c)
	}
	n := e.batch.rows
	if n == 0 {
		return nil
	}
	if err := e.batch.upload(); err != nil {
		return err
	}
	var selA, selB unsafe.Pointer
	if err := phErr(C.ph_dev_alloc(e.ctx, C.int64_t(n*4), &selA)); err != nil {
		return err
	}
	defer C.ph_dev_free(e.ctx, selA)
	if err := phErr(C.ph_dev_alloc(e.ctx, C.int64_t(n*4), &selB)); err != nil {
		return err
	}
	defer C.ph_dev_free(e.ctx, selB)
	var cur *C.int32_t
	cnt := C.int64_t(n)
	for ci := range e.preds {
		if cnt == 0 {
			break
		}
		p := e.preds[ci]
		k := 0
		for i, c := range e.cols {
			if c == int(p.col) {
				k = i
			}
		}
		col := e.batch.dev[k]
		kc := p.k
		if col._type == C.PH_DEC64 && kc._type == C.PH_I32 { // integer literal against DECIMAL / HUGEINT: cast to the column's type
			kc._type = C.PH_DEC64
			kc.scale = 0
		}
		if col._type == C.PH_CODE8 && kc._type == C.PH_STR { // VARCHAR '=' on a dictionary column: literal -> code
			kc._type = C.PH_I32
			kc.i = C.int64_t(e.batch.codeOf(k, C.GoString(kc.s)))
		}
		out := (*C.int32_t)(selA)
		if ci&1 == 1 {
			out = (*C.int32_t)(selB)
		}
		var m C.int64_t
		if err := phErr(C.ph_filter_select(e.ctx, &col, C.int64_t(n), p.op, &kc, cur, cnt, out, &m)); err != nil {
			return err
		}
		cur, cnt = out, m
	}
	sel := make([]int32, int(cnt))
	if cnt > 0 && len(e.preds) > 0 {
		if err := phErr(C.ph_dev_download(e.ctx, unsafe.Pointer(&sel[0]), unsafe.Pointer(cur), cnt*4)); err != nil {
			return err
		}
	}
	// split the batch selection back into per-chunk selection vectors ([]int, select_vector.go:7-9)
	p, start := 0, 0
	for _, c := range chunks {
		if len(e.preds) == 0 {
			e.ready = append(e.ready, readyChunk{c: c, sel: nil, n: c.Card()})
			continue
		}
		sv := chunk.NewSelectVector(util.DefaultVectorSize)
		k := 0
		for p < len(sel) && int(sel[p]) < start+c.Card() {
			sv.SetIndex(k, int(sel[p])-start)
			k++
			p++
		}
		start += c.Card()
		if k > 0 {
			e.ready = append(e.ready, readyChunk{c: c, sel: sv, n: k})
		}
	}
	return nil
}

func (e *gpuFilterExecutor) Execute(input, output *chunk.Chunk) (OperatorResult, error) {
	ensureOutputChunk(e.op, output)
	for len(e.ready) == 0 && !e.childDone {
		if err := e.fill(); err != nil {
			return InvalidOpResult, err
		}
	}
	if len(e.ready) == 0 {
		return Done, nil
	}
	it := e.ready[0]
	e.ready = e.ready[1:]
	if it.sel == nil {
		output.Reference(it.c)
		return haveMoreOutput, nil
	}
	indice := make([]int, it.c.ColumnCount())
	for i := range indice {
		indice[i] = i
	}
	output.SliceIndice(it.c, it.sel, it.n, 0, indice)
	return haveMoreOutput, nil
}

func (e *gpuFilterExecutor) Close() error {
	if e.batch != nil {
		e.batch.close()
	}
	e.freeKeep()
	if e.ctx != nil {
		C.ph_ctx_destroy(e.ctx)
		e.ctx = nil
	}
	return e.child.Close()
}

// ---------------------------------------------------------------------------------- Join

// gpuJoinExecutor <- joinExecutor (executor_join.go:27-264): children[0] probes, children[1] is built
// (joinBuildHashTable :237-264). Keys are staged to the device; payload columns stay in the host
// chunks and are picked by the (probe row, build row) pairs the device returns — INNER; LEFT adds
// the unmatched probe rows with constant-NULL build columns (NextLeftJoin, join_scan.go:67-88);
// SEMI / ANTI slice the probe chunks by the found flag (NextSemiOrAntiJoin :102-120).
type gpuJoinExecutor struct {
	op          *PhysicalOperator
	probe       OperatorExec
	build       OperatorExec
	ctx         *C.ph_ctx
	joinTyp     LOT_JoinType
	probeKeys   []int
	buildKeys   []int
	probeBatch  *deviceBatch
	buildBatch  *deviceBatch
	buildChunks []*chunk.Chunk
	buildStart  []int
	join        *C.ph_join
	built       bool
	probeDone   bool
	ready       []*chunk.Chunk
	outputEx    *ExprExec
	nProbeCols  int
	nBuildCols  int
}

func tryNewGpuJoinExecutor(op *PhysicalOperator, cfg *util.Config, txn *storage.Txn, children []OperatorExec) (OperatorExec, error) {
	if !gpuEnabled || len(children) != 2 {
		return nil, errFallback
	}
	info, ok := op.Info.(*JoinOpInfo)
	if !ok {
		return nil, errFallback
	}
	e := &gpuJoinExecutor{op: op, probe: children[0], build: children[1], joinTyp: info.JoinTyp}
	switch info.JoinTyp {
	case LOT_JoinTypeInner, LOT_JoinTypeLeft, LOT_JoinTypeSEMI, LOT_JoinTypeANTI:
	default:
		return nil, errFallback
	}
	// equi-join conditions only: `left column = right column`
	for _, cond := range info.OnConds {
		if cond.Typ != ET_Func || cond.FuncName() != FuncEqual || len(cond.Children) != 2 {
			return nil, errFallback
		}
		lt, lc, lok := colRefOf(stripCast(cond.Children[0]))
		rt, rc, rok := colRefOf(stripCast(cond.Children[1]))
		if !lok || !rok {
			return nil, errFallback
		}
		if lt == -1 && rt == -2 {
			e.probeKeys, e.buildKeys = append(e.probeKeys, lc), append(e.buildKeys, rc)
		} else if lt == -2 && rt == -1 {
			e.probeKeys, e.buildKeys = append(e.probeKeys, rc), append(e.buildKeys, lc)
		} else {
			return nil, errFallback
		}
	}
	if len(e.probeKeys) == 0 || len(e.probeKeys) > 4 {
		return nil, errFallback
	}
	pt := make([]common.LType, len(op.Children[0].Outputs))
	for i, o := range op.Children[0].Outputs {
		pt[i] = o.DataTyp
	}
	bt := make([]common.LType, len(op.Children[1].Outputs))
	for i, o := range op.Children[1].Outputs {
		bt[i] = o.DataTyp
	}
	e.nProbeCols, e.nBuildCols = len(pt), len(bt)
	for i := range e.probeKeys { // VARCHAR keys and keys of different widths stay on the CPU executor
		a, b := pt[e.probeKeys[i]].GetInternalType(), bt[e.buildKeys[i]].GetInternalType()
		if a == common.VARCHAR || b == common.VARCHAR || a != b {
			return nil, errFallback
		}
	}
	var err error
	if e.ctx, err = newGpuCtx(); err != nil {
		return nil, err
	}
	if e.probeBatch, err = newDeviceBatch(e.ctx, pt, e.probeKeys, nil); err == nil {
		e.buildBatch, err = newDeviceBatch(e.ctx, bt, e.buildKeys, nil)
	}
	if err != nil {
		C.ph_ctx_destroy(e.ctx)
		return nil, err
	}
	return e, nil
}

func (e *gpuJoinExecutor) Init() error {
	e.outputEx = NewExprExec(e.op.Outputs...) // evalJoinOutput (executor_join.go:209-235)
	return nil
}

func (e *gpuJoinExecutor) buildTable() error {
	total := 0
	for {
		c := &chunk.Chunk{}
		res, err := e.build.Execute(nil, c)
		if err != nil {
			return err
		}
		if res == Done {
			break
		}
		if c.Card() == 0 {
			continue
		}
		if err = e.buildBatch.append(c); err != nil {
			return err
		}
		e.buildStart = append(e.buildStart, total)
		e.buildChunks = append(e.buildChunks, c)
		total += c.Card()
	}
	if err := e.buildBatch.upload(); err != nil {
		return err
	}
	// One INTEGER / BIGINT key: hand the library the key range seen while staging. A dense range (a
	// primary key: customer, supplier, orders) then builds a direct table addressed by key - lo instead
	// of a hash table; anything else falls back inside the library to what ph_join_build builds.
	flags, lo, hi := C.int32_t(0), C.int64_t(0), C.int64_t(0)
	if len(e.buildKeys) == 1 && e.buildBatch.sc[0].ranged {
		flags, lo, hi = C.PH_JOIN_KEY_RANGE, C.int64_t(e.buildBatch.sc[0].lo), C.int64_t(e.buildBatch.sc[0].hi)
	}
	return phErr(C.ph_join_build_ex(e.ctx, &e.buildBatch.dev[0], C.int32_t(len(e.buildKeys)), nil, C.int64_t(total), flags, lo, hi, &e.join))
}

func chunkOf(starts []int, row int) int { // last chunk whose first row <= row
	lo, hi := 0, len(starts)
	for lo+1 < hi {
		mid := (lo + hi) / 2
		if starts[mid] <= row {
			lo = mid
		} else {
			hi = mid
		}
	}
	return lo
}

func (e *gpuJoinExecutor) emit(left *chunk.Chunk, right *chunk.Chunk) error {
	out := &chunk.Chunk{}
	ensureOutputChunk(e.op, out)
	if err := e.outputEx.executeExprs([]*chunk.Chunk{left, right, nil}, out); err != nil {
		return err
	}
	if out.Card() > 0 {
		e.ready = append(e.ready, out)
	}
	return nil
}

func (e *gpuJoinExecutor) probeBatchOnce() error {
	var chunks []*chunk.Chunk
	var starts []int
	e.probeBatch.reset()
	for len(chunks) < gpuBatchChunks {
		c := &chunk.Chunk{}
		res, err := e.probe.Execute(nil, c)
		if err != nil {
			return err
		}
		if res == Done {
			e.probeDone = true
			break
		}
		if c.Card() == 0 {
			continue
		}
		starts = append(starts, e.probeBatch.rows)
		if err = e.probeBatch.append(c); err != nil {
			return err
		}
		chunks = append(chunks, c)
	}
	n := e.probeBatch.rows
	if n == 0 {
		return nil
	}
	if err := e.probeBatch.upload(); err != nil {
		return err
	}
	probeTypes := func(c *chunk.Chunk) []common.LType {
		t := make([]common.LType, c.ColumnCount())
		for i, v := range c.Data {
			t[i] = v.Typ()
		}
		return t
	}
	if e.joinTyp == LOT_JoinTypeSEMI || e.joinTyp == LOT_JoinTypeANTI {
		var fd unsafe.Pointer
		if err := phErr(C.ph_dev_alloc(e.ctx, C.int64_t(n), &fd)); err != nil {
			return err
		}
		defer C.ph_dev_free(e.ctx, fd)
		found := make([]uint8, n)
		if err := phErr(C.ph_join_probe_mark(e.join, &e.probeBatch.dev[0], nil, C.int64_t(n), (*C.uint8_t)(fd))); err != nil {
			return err
		}
		if err := phErr(C.ph_dev_download(e.ctx, unsafe.Pointer(&found[0]), fd, C.int64_t(n))); err != nil {
			return err
		}
		want := uint8(1)
		if e.joinTyp == LOT_JoinTypeANTI {
			want = 0
		}
		for ci, c := range chunks {
			sv := chunk.NewSelectVector(util.DefaultVectorSize)
			k := 0
			for i := 0; i < c.Card(); i++ {
				if found[starts[ci]+i] == want {
					sv.SetIndex(k, i)
					k++
				}
			}
			if k == 0 {
				continue
			}
			left := &chunk.Chunk{}
			left.Init(probeTypes(c), util.DefaultVectorSize)
			left.Slice(c, sv, k, 0)
			if err := e.emit(left, nil); err != nil {
				return err
			}
		}
		return nil
	}
	capPairs := C.int64_t(n + 1024)
	var op, ob unsafe.Pointer
	var m C.int64_t
	for attempt := 0; ; attempt++ {
		if err := phErr(C.ph_dev_alloc(e.ctx, capPairs*4, &op)); err != nil {
			return err
		}
		if err := phErr(C.ph_dev_alloc(e.ctx, capPairs*4, &ob)); err != nil {
			return err
		}
		rc := C.ph_join_probe_inner(e.join, &e.probeBatch.dev[0], nil, C.int64_t(n), (*C.int32_t)(op), (*C.int32_t)(ob), capPairs, &m)
		if rc == C.PH_OK {
			break
		}
		C.ph_dev_free(e.ctx, op)
		C.ph_dev_free(e.ctx, ob)
		if rc != C.PH_ECAPACITY || attempt == 1 {
			return phErr(rc)
		}
		capPairs = m // duplicate build keys: retry with the exact size
	}
	defer C.ph_dev_free(e.ctx, op)
	defer C.ph_dev_free(e.ctx, ob)
	pr, br := make([]int32, int(m)+1), make([]int32, int(m)+1)
	if m > 0 {
		if err := phErr(C.ph_dev_download(e.ctx, unsafe.Pointer(&pr[0]), op, m*4)); err != nil {
			return err
		}
		if err := phErr(C.ph_dev_download(e.ctx, unsafe.Pointer(&br[0]), ob, m*4)); err != nil {
			return err
		}
	}
	// <= 2048 pairs at a time: both sides become DICT views over their source chunk when all pairs
	// of the slice come from one chunk each (the common case: pairs arrive in probe order), else
	// the rows are copied cell by cell (gatherResult, join_scan.go:250-278)
	buildTypes := make([]common.LType, e.nBuildCols)
	for i, o := range e.op.Children[1].Outputs {
		buildTypes[i] = o.DataTyp
	}
	matched := make([]bool, n)
	for i := 0; i < int(m); {
		pc := chunkOf(starts, int(pr[i]))
		bc := chunkOf(e.buildStart, int(br[i]))
		ls, rs := chunk.NewSelectVector(util.DefaultVectorSize), chunk.NewSelectVector(util.DefaultVectorSize)
		k := 0
		for i < int(m) && k < util.DefaultVectorSize && chunkOf(starts, int(pr[i])) == pc && chunkOf(e.buildStart, int(br[i])) == bc {
			ls.SetIndex(k, int(pr[i])-starts[pc])
			rs.SetIndex(k, int(br[i])-e.buildStart[bc])
			matched[pr[i]] = true
			k++
			i++
		}
		left, right := &chunk.Chunk{}, &chunk.Chunk{}
		left.Init(probeTypes(chunks[pc]), util.DefaultVectorSize)
		left.Slice(chunks[pc], ls, k, 0)
		right.Init(buildTypes, util.DefaultVectorSize)
		right.Slice(e.buildChunks[bc], rs, k, 0)
		if err := e.emit(left, right); err != nil {
			return err
		}
	}
	if e.joinTyp == LOT_JoinTypeLeft {
		for ci, c := range chunks {
			sv := chunk.NewSelectVector(util.DefaultVectorSize)
			k := 0
			for i := 0; i < c.Card(); i++ {
				if !matched[starts[ci]+i] {
					sv.SetIndex(k, i)
					k++
				}
			}
			if k == 0 {
				continue
			}
			left, right := &chunk.Chunk{}, &chunk.Chunk{}
			left.Init(probeTypes(c), util.DefaultVectorSize)
			left.Slice(c, sv, k, 0)
			right.Init(buildTypes, util.DefaultVectorSize)
			for _, v := range right.Data { // every build-side column a constant NULL
				v.SetPhyFormat(chunk.PF_CONST)
				chunk.SetNullInPhyFormatConst(v, true)
			}
			right.SetCard(k)
			if err := e.emit(left, right); err != nil {
				return err
			}
		}
	}
	return nil
}

func (e *gpuJoinExecutor) Execute(input, output *chunk.Chunk) (OperatorResult, error) {
	if !e.built {
		if err := e.buildTable(); err != nil {
			return InvalidOpResult, err
		}
		e.built = true
	}
	for len(e.ready) == 0 && !e.probeDone {
		if int64(C.ph_join_count(e.join)) == 0 && e.joinTyp != LOT_JoinTypeANTI && e.joinTyp != LOT_JoinTypeLeft {
			e.probeDone = true // empty build side: no inner / semi row can match
			break
		}
		if err := e.probeBatchOnce(); err != nil {
			return InvalidOpResult, err
		}
	}
	if len(e.ready) == 0 {
		return Done, nil
	}
	output.Reference(e.ready[0])
	e.ready = e.ready[1:]
	return haveMoreOutput, nil
}

func (e *gpuJoinExecutor) Close() error {
	if e.join != nil {
		C.ph_join_free(e.join)
		e.join = nil
	}
	if e.probeBatch != nil {
		e.probeBatch.close()
	}
	if e.buildBatch != nil {
		e.buildBatch.close()
	}
	if e.ctx != nil {
		C.ph_ctx_destroy(e.ctx)
		e.ctx = nil
	}
	e.buildChunks, e.ready = nil, nil
	if err := e.probe.Close(); err != nil {
		return err
	}
	return e.build.Close()
}

// ---------------------------------------------------------------------------------- multi-GPU

// gpuComm wraps the exchange entry points for a coordinator that runs one goroutine (locked to an
// OS thread) per GPU: rank 0 calls NewGpuCommID and ships the 128 bytes to the other ranks over the
// coordinator's own channel; every rank then calls newGpuComm. A partitioned join stage is
//   ph_partition_dev -> ph_gather per column -> exchangeCounts (the stage's one host round trip)
//   -> exchangeColumns (one grouped RCCL all-to-all over xGMI) -> local ph_join_build / probe.
type gpuComm struct{ h *C.ph_comm }

func NewGpuCommID() ([]byte, error) {
	id := make([]byte, C.PH_COMM_ID_BYTES)
	if err := phErr(C.ph_comm_unique_id(unsafe.Pointer(&id[0]))); err != nil {
		return nil, err
	}
	return id, nil
}

func newGpuComm(ctx *C.ph_ctx, nranks, rank int, id []byte) (*gpuComm, error) {
	c := &gpuComm{}
	if err := phErr(C.ph_comm_init(ctx, C.int32_t(nranks), C.int32_t(rank), unsafe.Pointer(&id[0]), &c.h)); err != nil {
		return nil, err
	}
	return c, nil
}

func (c *gpuComm) close() { C.ph_comm_destroy(c.h) }

// exchangeCounts: this rank's per-destination counts (device, from ph_partition_dev) -> the
// nranks x nranks matrix on every rank
func (c *gpuComm) exchangeCounts(countsDev unsafe.Pointer) ([]int64, error) {
	n := int(C.ph_comm_nranks(c.h))
	m := make([]int64, n*n)
	if err := phErr(C.ph_comm_exchange_counts(c.h, (*C.int64_t)(countsDev), (*C.int64_t)(unsafe.Pointer(&m[0])))); err != nil {
		return nil, err
	}
	return m, nil
}

// exchangeColumns: all-to-all of column buffers ordered by destination; returns rows received
func (c *gpuComm) exchangeColumns(ctx *C.ph_ctx, send []unsafe.Pointer, widths []int32, matrix []int64) ([]unsafe.Pointer, int64, error) {
	n := int(C.ph_comm_nranks(c.h))
	so, ro := make([]int64, n+1), make([]int64, n+1)
	if err := phErr(C.ph_exchange_layout((*C.int64_t)(unsafe.Pointer(&matrix[0])), C.int32_t(n), C.ph_comm_rank(c.h),
		(*C.int64_t)(unsafe.Pointer(&so[0])), (*C.int64_t)(unsafe.Pointer(&ro[0])))); err != nil {
		return nil, 0, err
	}
	recv := make([]unsafe.Pointer, len(send))
	for k := range send {
		if err := phErr(C.ph_dev_alloc(ctx, C.int64_t(ro[n]*int64(widths[k])+64), &recv[k])); err != nil {
			return nil, 0, err
		}
	}
	// pointer arrays live in C memory for the call (cgo: no Go pointers to Go pointers)
	sp := (*[1 << 20]unsafe.Pointer)(C.malloc(C.size_t(len(send)) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	rp := (*[1 << 20]unsafe.Pointer)(C.malloc(C.size_t(len(send)) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	defer C.free(unsafe.Pointer(sp))
	defer C.free(unsafe.Pointer(rp))
	for k := range send {
		sp[k], rp[k] = send[k], recv[k]
	}
	err := phErr(C.ph_comm_exchange_columns(c.h, C.int32_t(len(send)), (*unsafe.Pointer)(unsafe.Pointer(sp)), (*unsafe.Pointer)(unsafe.Pointer(rp)),
		(*C.int32_t)(unsafe.Pointer(&widths[0])), (*C.int64_t)(unsafe.Pointer(&matrix[0]))))
	return recv, ro[n], err
}

// silence "imported and not used" for helpers only some build tags use
var _ = fmt.Sprintf
