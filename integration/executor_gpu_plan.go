// executor_gpu_plan.go — gpuResidentPlanExecutor: a whole operator SUBTREE over resident tables behind one
// OperatorExec, handed to the library as a ph_plan (include/planhip.h, "resident plans").
//
// Placement: pkg/compute/executor_gpu_plan.go, next to executor_gpu.go (same package, same cgo preamble rules).
// The Go twin of plan_amd/csrc/host/executors.cpp's ResidentPlan / gpuResidentPlanExecutor, which IS compiled
// and tested (tests/test_host_layer.py: `host_tester q3|q9 <sf> resident` and `host_tester tpch <id>` reproduce
// cases/tpch/1g/plan/q{1,3,4,5,6,9,12,14,19}.txt byte for byte). NOT COMPILED HERE (no Go toolchain, SURVEY.md §8c).
//
// What it does. buildOperatorExec (executor.go:305-350) builds executors bottom-up. When it reaches a POT_Agg whose
// subtree consists of POT_Join / POT_Filter / POT_Project nodes over POT_Scan leaves of tables that are (or can be
// made) resident, the Agg case arm calls tryNewGpuResidentPlanExecutor first (INTEGRATION.md §2): the subtree is
// written down as a flat ph_plan_node array — each node with exactly the fields PhysicalOperator carries for it:
//   Scan    ScanOpInfo's table + the pruned Outputs + Filters (pushed-down conjuncts)
//   Filter  Filters
//   Join    JoinOpInfo.JoinTyp + OnConds (children[0] probes, children[1] is built: executor_join.go:237-264,
//           cond.Children[0] is the probe-side key, Children[1] the build-side key: join_types.go:49-98) + Outputs
//   Project Projects
//   Agg     AggOpInfo.GroupBys / Aggs (+ Filters = HAVING and Outputs, which stay with the reference's ExprExec)
// and the library runs it: every physical choice (direct / gated / node / chained tables, lookups vs pairs,
// semi-join marks, sideways information passing, merge and streaming forms, late materialisation) is made there
// from the resident tables' statistics (ph_table_col_range / ph_table_col_stats / ph_table_declare_unique) — the
// planner hints nothing and this file chooses nothing. Anything the lowering below does not recognise returns
// errFallback and buildOperatorExec keeps the per-operator executors (executor_gpu.go) or the CPU ones.
package compute

/*
#include <stdlib.h>
#include <string.h>
#include "planhip.h"
*/
import "C"

import (
	"math"
	"strings"
	"unsafe"

	"github.com/daviszhen/plan/pkg/chunk"
	"github.com/daviszhen/plan/pkg/common"
	"github.com/daviszhen/plan/pkg/storage"
	"github.com/daviszhen/plan/pkg/util"
)

// cArena owns the C memory a descriptor points into until ph_plan_create has copied it.
type cArena struct{ ptrs []unsafe.Pointer }

func (a *cArena) alloc(n uintptr) unsafe.Pointer {
	if n == 0 {
		n = 8
	}
	p := C.calloc(1, C.size_t(n))
	a.ptrs = append(a.ptrs, p)
	return p
}
func (a *cArena) i32s(v []int) *C.int32_t {
	p := (*C.int32_t)(a.alloc(uintptr(len(v)) * 4))
	s := unsafe.Slice(p, len(v)+1)
	for i, x := range v {
		s[i] = C.int32_t(x)
	}
	return p
}
func (a *cArena) cstr(s string) *C.char {
	p := C.CString(s)
	a.ptrs = append(a.ptrs, unsafe.Pointer(p))
	return p
}
func (a *cArena) free() {
	for _, p := range a.ptrs {
		C.free(p)
	}
	a.ptrs = nil
}

// planBuilder accumulates the flat node array; every lowered node remembers, per output column, the LType and the
// resident column it is an unchanged copy of (dictionary codes need their dictionary when they come back as keys).
type planNodeMeta struct {
	types []common.LType
	dicts [][]string
}

type planBuilder struct {
	cfg    *util.Config
	txn    *storage.Txn
	arena  cArena
	nodes  []C.ph_plan_node
	meta   []planNodeMeta
	tables []*residentTable
}

func (b *planBuilder) add(n C.ph_plan_node, m planNodeMeta) int {
	b.nodes = append(b.nodes, n)
	b.meta = append(b.meta, m)
	return len(b.nodes) - 1
}

// position of a column reference in the concatenation [children[0] columns | children[1] columns]
// (ColumnBind.table(): -1 = children[0], -2 = children[1]; executeColumnRef, expr_exec.go:248-265)
func childColumn(e *Expr, nLeft int) (int, bool) {
	e = stripCast(e)
	tab, col, ok := colRefOf(e)
	if !ok {
		return 0, false
	}
	switch tab {
	case -1:
		return col, true
	case -2:
		return nLeft + col, true
	}
	return 0, false
}

// ---- boolean trees: what executeSelect walks (execSelectAnd / Or / Compare, expr_exec.go:342-530)

// conjuncts of the form `column OP literal` go to ph_pred (a build or a probe can absorb those); everything else —
// OR / IN lists, column-vs-column comparisons — becomes ONE ph_bool tree AND-ed behind them
func (b *planBuilder) lowerFilters(filters []*Expr, nLeft int) (preds []C.ph_pred, bools []C.ph_bool, ok bool) {
	var rest []*Expr
	var flat func(e *Expr)
	flat = func(e *Expr) {
		if e != nil && e.Typ == ET_Func && e.FuncName() == FuncAnd {
			for _, c := range e.Children {
				flat(c)
			}
			return
		}
		rest = append(rest, e)
	}
	for _, f := range filters {
		flat(f)
	}
	var complex []*Expr
	for _, e := range rest {
		if p, simple := b.lowerSimple(e, nLeft); simple {
			preds = append(preds, p)
		} else {
			complex = append(complex, e)
		}
	}
	if len(complex) == 0 {
		return preds, nil, true
	}
	root := &Expr{Typ: ET_Func, Children: complex}
	bools = make([]C.ph_bool, 1)
	if !b.flattenBool(root, true, 0, nLeft, &bools) {
		return nil, nil, false
	}
	return preds, bools, true
}

func (b *planBuilder) lowerSimple(e *Expr, nLeft int) (C.ph_pred, bool) {
	var p C.ph_pred
	if e == nil || e.Typ != ET_Func || len(e.Children) != 2 {
		return p, false
	}
	op, isCmp := cmpOps[e.FuncName()]
	if !isCmp {
		return p, false
	}
	col, ok := childColumn(e.Children[0], nLeft)
	if !ok {
		return p, false
	}
	k, ok := lowerConst(e.Children[1], &b.arena.ptrs)
	if !ok {
		return p, false
	}
	p.col, p.op, p.k = C.int32_t(col), op, k
	return p, true
}

// node `at` of the flat array <- expression e; the children of AND / OR are appended contiguously. isAndRoot: the
// synthetic root that AND-s the complex conjuncts.
func (b *planBuilder) flattenBool(e *Expr, isAndRoot bool, at int, nLeft int, out *[]C.ph_bool) bool {
	var n C.ph_bool
	name := ""
	if !isAndRoot {
		if e == nil || e.Typ != ET_Func {
			return false
		}
		name = e.FuncName()
	}
	kids := e.Children
	switch {
	case isAndRoot || name == FuncAnd || name == FuncOr:
		n.kind = C.PH_B_AND
		if name == FuncOr {
			n.kind = C.PH_B_OR
		}
	case name == FuncIn && len(e.Children) >= 2:
		// a IN (x, y, ..) = in(a,x) OR in(a,y) ..; `in` selects like `=` (function_operator_boolean.go:419-429)
		n.kind = C.PH_B_OR
		kids = nil
		for _, v := range e.Children[1:] {
			kids = append(kids, &Expr{Typ: ET_Func, Info: eqFuncInfo(), Children: []*Expr{e.Children[0], v}})
		}
	default:
		op, isCmp := cmpOps[name]
		if !isCmp || len(e.Children) != 2 {
			return false
		}
		col, ok := childColumn(e.Children[0], nLeft)
		if !ok {
			return false
		}
		n.kind, n.col, n.op = C.PH_B_CMP, C.int32_t(col), op
		if col2, isCol := childColumn(e.Children[1], nLeft); isCol { // column OP column (l_commitdate < l_receiptdate)
			n.k._type = C.PH_COLREF
			n.k.i = C.int64_t(col2)
		} else {
			k, ok := lowerConst(e.Children[1], &b.arena.ptrs)
			if !ok {
				return false
			}
			n.k = k
		}
		(*out)[at] = n
		return true
	}
	n.first_child = C.int32_t(len(*out))
	n.nchildren = C.int32_t(len(kids))
	(*out)[at] = n
	base := len(*out)
	*out = append(*out, make([]C.ph_bool, len(kids))...)
	for i, c := range kids {
		if !b.flattenBool(c, false, base+i, nLeft, out) {
			return false
		}
	}
	return true
}

// the FunctionInfo of `=` for the comparisons an IN list expands to (only FuncName() is looked at here)
func eqFuncInfo() *FunctionInfo { return &FunctionInfo{FunImpl: &Function{_name: FuncEqual}} }

// ---- expressions over a child's output columns

func (b *planBuilder) lowerPlanExpr(e *Expr, nLeft int) (C.ph_plan_expr, bool) {
	var x C.ph_plan_expr
	e = stripCast(e)
	if e == nil {
		return x, false
	}
	if col, ok := childColumn(e, nLeft); ok {
		x.kind, x.col = C.PH_PE_COL, C.int32_t(col)
		return x, true
	}
	if e.Typ == ET_Func && e.FuncName() == FuncExtract && len(e.Children) == 2 {
		// extract(year from <DATE column>): Children[0] names the part, Children[1] is the date (function_scalar.go:1509-1563)
		part := stripCast(e.Children[0])
		col, ok := childColumn(e.Children[1], nLeft)
		if ok && part.Typ == ET_Const && strings.EqualFold(part.ConstValue.String, "year") {
			x.kind, x.col = C.PH_PE_YEAR, C.int32_t(col)
			return x, true
		}
		return x, false
	}
	if e.Typ == ET_Func && e.FuncName() == FuncSubstring && len(e.Children) == 3 {
		// substring(<VARCHAR column> FROM <const> FOR <const>) (substringFunc, function_operator_binary.go:553-625): computed inside the plan
		// as string codes (PH_PE_SUBSTR); a filter's = / <> / IN against VARCHAR constants and a group key read it
		col, ok := childColumn(e.Children[0], nLeft)
		from, length := stripCast(e.Children[1]), stripCast(e.Children[2])
		if ok && from.Typ == ET_Const && length.Typ == ET_Const {
			x.kind, x.col = C.PH_PE_SUBSTR, C.int32_t(col)
			x.sub_offset, x.sub_length = C.int64_t(from.ConstValue.Integer), C.int64_t(length.ConstValue.Integer)
			return x, true
		}
		return x, false
	}
	if e.Typ == ET_Func && e.FuncName() == FuncCase && len(e.Children) == 3 {
		// Children[0] = ELSE, then (WHEN, THEN) pairs (executeCase, expr_exec.go:144-246); one WHEN here
		var thenP, elseP []C.ph_rpn
		if !b.lowerRPNChild(e.Children[2], nLeft, &thenP) || !b.lowerRPNChild(e.Children[0], nLeft, &elseP) || len(thenP) > 12 || len(elseP) > 12 {
			return x, false
		}
		when := make([]C.ph_bool, 1)
		if !b.flattenBool(e.Children[1], false, 0, nLeft, &when) {
			return x, false
		}
		x.kind, x.col = C.PH_PE_CASE, -1
		x.nprog, x.nelse = C.int32_t(len(thenP)), C.int32_t(len(elseP))
		for i := range thenP {
			x.prog[i] = thenP[i]
		}
		for i := range elseP {
			x.else_prog[i] = elseP[i]
		}
		wp := (*C.ph_bool)(b.arena.alloc(uintptr(len(when)) * unsafe.Sizeof(when[0])))
		copy(unsafe.Slice(wp, len(when)), when)
		x.nwhen, x.when = C.int32_t(len(when)), wp
		if e.DataTyp.Id == common.LTID_INTEGER { // THEN 1 ELSE 0: an INTEGER result, summed into a HUGEINT
			x.result_int = 1
		}
		return x, true
	}
	var prog []C.ph_rpn
	if !b.lowerRPNChild(e, nLeft, &prog) || len(prog) > 12 {
		return x, false
	}
	x.kind, x.col, x.nprog = C.PH_PE_DECIMAL, -1, C.int32_t(len(prog))
	for i := range prog {
		x.prog[i] = prog[i]
	}
	return x, true
}

// lowerRPN of executor_gpu.go addresses ONE child's columns; here a column may come from either child of a join
func (b *planBuilder) lowerRPNChild(e *Expr, nLeft int, out *[]C.ph_rpn) bool {
	e = stripCast(e)
	if e == nil {
		return false
	}
	if col, ok := childColumn(e, nLeft); ok {
		*out = append(*out, C.ph_rpn{op: C.PH_X_COL, col: C.int32_t(col)})
		return true
	}
	switch e.Typ {
	case ET_Const:
		var tmp []C.ph_rpn
		if !lowerRPN(e, -1, &tmp) {
			return false
		}
		*out = append(*out, tmp...)
		return true
	case ET_Func:
		ops := map[string]C.int32_t{FuncAdd: C.PH_X_ADD, FuncSubtract: C.PH_X_SUB, FuncMultiply: C.PH_X_MUL}
		op, ok := ops[e.FuncName()]
		if !ok || len(e.Children) != 2 || !b.lowerRPNChild(e.Children[0], nLeft, out) || !b.lowerRPNChild(e.Children[1], nLeft, out) {
			return false
		}
		*out = append(*out, C.ph_rpn{op: op, col: -1})
		return true
	}
	return false
}

// A FLOAT / DOUBLE expression (the binder typed it so: a FLOAT literal beside a DECIMAL or INTEGER operand, avg(INTEGER) beside anything)
// as a PH_PE_FLOAT program: column references keep the cast the binder put on them out of the program (ph_float_eval casts by column type, as
// tryCast*ToFloat32 / ..Float64 do), FLOAT literals carry their float32 bits, the arithmetic and — as the last step — one comparison follow.
// wide: the expression's type is DOUBLE. A Filter `l_quantity < 0.2 * avg` lowers to Project(flag) + Filter(flag = 1): see lower(POT_Filter).
func (b *planBuilder) lowerFloatRPN(e *Expr, nLeft int, out *[]C.ph_rpn) bool {
	e = stripCast(e)
	if e == nil {
		return false
	}
	if col, ok := childColumn(e, nLeft); ok {
		*out = append(*out, C.ph_rpn{op: C.PH_X_COL, col: C.int32_t(col)})
		return true
	}
	switch e.Typ {
	case ET_Const:
		if e.ConstValue.Type != ConstTypeFloat {
			return false
		}
		bits := math.Float32bits(float32(e.ConstValue.Float))
		*out = append(*out, C.ph_rpn{op: C.PH_X_CONST, col: -1, ival: C.int64_t(bits)})
		return true
	case ET_Func:
		ops := map[string]C.int32_t{FuncAdd: C.PH_X_ADD, FuncSubtract: C.PH_X_SUB, FuncMultiply: C.PH_X_MUL, FuncDivide: C.PH_X_DIV,
			FuncLess: C.PH_X_LT, FuncLessEqual: C.PH_X_LE, FuncGreater: C.PH_X_GT, FuncGreaterEqual: C.PH_X_GE}
		op, ok := ops[e.FuncName()]
		if !ok || len(e.Children) != 2 || !b.lowerFloatRPN(e.Children[0], nLeft, out) || !b.lowerFloatRPN(e.Children[1], nLeft, out) {
			return false
		}
		*out = append(*out, C.ph_rpn{op: op, col: -1})
		return true
	}
	return false
}

func (b *planBuilder) lowerFloatExpr(e *Expr, nLeft int, truth bool) (C.ph_plan_expr, bool) {
	var x C.ph_plan_expr
	var prog []C.ph_rpn
	if !b.lowerFloatRPN(e, nLeft, &prog) || len(prog) > 12 {
		return x, false
	}
	x.kind, x.col, x.nprog = C.PH_PE_FLOAT, -1, C.int32_t(len(prog))
	for i := range prog {
		x.prog[i] = prog[i]
	}
	if truth {
		x.result_int = 1
	}
	if stripCast(e.Children[0]).DataTyp.Id == common.LTID_DOUBLE || stripCast(e.Children[1]).DataTyp.Id == common.LTID_DOUBLE {
		x.float_wide = 1
	}
	return x, true
}

// ---- the subtree, bottom-up

func (b *planBuilder) lower(op *PhysicalOperator) (int, bool) {
	switch op.Typ {
	case POT_Scan:
		if op.getScanTyp() != ScanTypeTable {
			return 0, false
		}
		rt, err := residentTableFor(op, b.cfg, b.txn) // loads the pruned columns once; declares the catalog's PRIMARY KEY
		if err != nil {
			return 0, false
		}
		b.tables = append(b.tables, rt)
		var n C.ph_plan_node
		n.kind = C.PH_PN_SCAN
		n.child[0], n.child[1] = -1, -1
		n.table = rt.h
		cols := make([]int, len(rt.types))
		for i := range cols {
			cols[i] = i // the resident table holds exactly the scan's pruned outputs, in order
		}
		n.ncols, n.cols = C.int32_t(len(cols)), b.arena.i32s(cols)
		preds, bools, ok := b.lowerFilters(op.Filters, 0)
		if !ok {
			return 0, false
		}
		b.setPreds(&n, preds, bools)
		return b.add(n, planNodeMeta{types: rt.types, dicts: rt.dicts}), true

	case POT_Filter:
		c, ok := b.lower(op.Children[0])
		if !ok {
			return 0, false
		}
		preds, bools, ok := b.lowerFilters(op.Filters, 0)
		if !ok {
			return 0, false
		}
		var n C.ph_plan_node
		n.kind = C.PH_PN_FILTER
		n.child[0], n.child[1] = C.int32_t(c), -1
		b.setPreds(&n, preds, bools)
		return b.add(n, b.meta[c]), true

	case POT_Join:
		var jt C.int32_t
		switch op.getJoinTyp() {
		case LOT_JoinTypeInner:
			jt = C.PH_JT_INNER
		case LOT_JoinTypeSEMI:
			jt = C.PH_JT_SEMI
		case LOT_JoinTypeANTI:
			jt = C.PH_JT_ANTI
		case LOT_JoinTypeLeft:
			jt = C.PH_JT_LEFT // NextLeftJoin inside the plan: the build side's columns carry a validity bitmap from here on
		default:
			return 0, false // MARK / cross: the per-operator executors
		}
		l, ok := b.lower(op.Children[0])
		if !ok {
			return 0, false
		}
		r, ok := b.lower(op.Children[1])
		if !ok {
			return 0, false
		}
		nLeft := len(b.meta[l].types)
		var pk, bk []int
		var residual []*Expr // the ON clause's non-equi conjuncts (Q21: l2.l_suppkey <> l1.l_suppkey): the join node's residual condition
		for _, cond := range op.getOnConds() {
			// equality of a probe-side and a build-side column: a key pair; anything else: residual
			if cond.Typ == ET_Func && cond.FuncName() == FuncEqual && len(cond.Children) == 2 {
				lc, ok1 := childColumn(cond.Children[0], nLeft)
				rc, ok2 := childColumn(cond.Children[1], nLeft)
				if ok1 && ok2 && lc < nLeft && rc >= nLeft {
					pk = append(pk, lc)
					bk = append(bk, rc-nLeft)
					continue
				}
			}
			residual = append(residual, cond)
		}
		if len(pk) == 0 || (len(residual) > 0 && jt == C.PH_JT_LEFT) {
			return 0, false
		}
		var residualBools []C.ph_bool
		if len(residual) > 0 { // one tree over [probe columns | build columns]: the column numbering the ON clause already uses
			root := &Expr{Typ: ET_Func, Children: residual}
			residualBools = make([]C.ph_bool, 1)
			if !b.flattenBool(root, true, 0, nLeft, &residualBools) {
				return 0, false
			}
		}
		all := planNodeMeta{types: append(append([]common.LType{}, b.meta[l].types...), b.meta[r].types...),
			dicts: append(append([][]string{}, b.meta[l].dicts...), b.meta[r].dicts...)}
		var out []int
		var m planNodeMeta
		for _, o := range op.Outputs { // evalJoinOutput (executor_join.go:209-235): column references only
			c, ok := childColumn(o, nLeft)
			if !ok {
				return 0, false
			}
			out = append(out, c)
			m.types = append(m.types, all.types[c])
			m.dicts = append(m.dicts, all.dicts[c])
		}
		var n C.ph_plan_node
		n.kind = C.PH_PN_JOIN
		n.child[0], n.child[1] = C.int32_t(l), C.int32_t(r)
		n.join_type = jt
		n.nkeys = C.int32_t(len(pk))
		n.probe_keys, n.build_keys = b.arena.i32s(pk), b.arena.i32s(bk)
		n.nout, n.out = C.int32_t(len(out)), b.arena.i32s(out)
		b.setPreds(&n, nil, residualBools)
		return b.add(n, m), true

	case POT_Project:
		c, ok := b.lower(op.Children[0])
		if !ok {
			return 0, false
		}
		ex := (*C.ph_plan_expr)(b.arena.alloc(uintptr(len(op.Projects)) * unsafe.Sizeof(C.ph_plan_expr{})))
		exs := unsafe.Slice(ex, len(op.Projects))
		var m planNodeMeta
		for i, p := range op.Projects {
			x, ok := b.lowerPlanExpr(p, 0)
			if !ok {
				return 0, false
			}
			exs[i] = x
			m.types = append(m.types, p.DataTyp)
			if x.kind == C.PH_PE_COL {
				m.dicts = append(m.dicts, b.meta[c].dicts[int(x.col)])
			} else {
				m.dicts = append(m.dicts, nil)
			}
		}
		var n C.ph_plan_node
		n.kind = C.PH_PN_PROJECT
		n.child[0], n.child[1] = C.int32_t(c), -1
		n.nexprs, n.exprs = C.int32_t(len(op.Projects)), ex
		return b.add(n, m), true

	case POT_Agg:
		// an aggregate BELOW other operators (Q18's IN-subquery: GROUP BY l_orderkey HAVING sum(l_quantity) > k). Its output
		// columns are [group columns | aggregate results] with FinalizeStates' types (function_aggr.go:1330-1365); the groups
		// stay on the device (ph_agg_keys_dev / ph_agg_values_dev inside the library). A HAVING of the inner aggregate is the
		// Filter of THIS operator over those columns: its column references carry the aggregate's own binding tags, and
		// mapping them to positions is left to the maintainer who compiles this file — with a HAVING the subtree falls back.
		info, ok := op.Info.(*AggOpInfo)
		if !ok || len(op.Filters) > 0 || len(op.Children) != 1 {
			return 0, false
		}
		c, ok := b.lower(op.Children[0])
		if !ok {
			return 0, false
		}
		var m planNodeMeta
		groups := make([]C.ph_plan_expr, 0, len(info.GroupBys))
		for _, g := range info.GroupBys {
			x, ok := b.lowerPlanExpr(g, 0)
			if !ok {
				return 0, false
			}
			groups = append(groups, x)
			m.types = append(m.types, g.DataTyp)
			if x.kind == C.PH_PE_COL {
				m.dicts = append(m.dicts, b.meta[c].dicts[int(x.col)])
			} else {
				m.dicts = append(m.dicts, nil)
			}
		}
		aggs := make([]C.ph_plan_agg, 0, len(info.Aggs))
		for _, a := range info.Aggs {
			if a == nil || a.Typ != ET_Func || a.GetFuncInfo().FunImpl == nil {
				return 0, false
			}
			kind, ok := aggKinds[strings.ToLower(a.FuncName())]
			if !ok {
				return 0, false
			}
			if a.GetFuncInfo().FunImpl.IsDistinct() { // SinkDistinctGrouping's side table lives inside the library for count(distinct x)
				if kind != C.PH_A_COUNT || len(a.Children) != 1 {
					return 0, false
				}
				kind = C.PH_A_COUNT_DISTINCT
			}
			var pa C.ph_plan_agg
			if kind == C.PH_A_COUNT && (len(a.Children) == 0 || stripCast(a.Children[0]).Typ == ET_Const) {
				pa.kind = C.PH_A_COUNT_STAR
			} else {
				if len(a.Children) != 1 {
					return 0, false
				}
				x, ok := b.lowerPlanExpr(a.Children[0], 0)
				if !ok {
					return 0, false
				}
				pa.kind, pa.arg = kind, x
			}
			aggs = append(aggs, pa)
			m.types = append(m.types, a.DataTyp) // the binder already typed the result as FinalizeStates produces it
			m.dicts = append(m.dicts, nil)
		}
		var n C.ph_plan_node
		n.kind = C.PH_PN_AGG
		n.child[0], n.child[1] = C.int32_t(c), -1
		if len(groups) > 0 {
			gp := (*C.ph_plan_expr)(b.arena.alloc(uintptr(len(groups)) * unsafe.Sizeof(groups[0])))
			copy(unsafe.Slice(gp, len(groups)), groups)
			n.ngroups, n.groups = C.int32_t(len(groups)), gp
		}
		if len(aggs) > 0 {
			ap := (*C.ph_plan_agg)(b.arena.alloc(uintptr(len(aggs)) * unsafe.Sizeof(aggs[0])))
			copy(unsafe.Slice(ap, len(aggs)), aggs)
			n.naggs, n.aggs = C.int32_t(len(aggs)), ap
		}
		return b.add(n, m), true
	}
	return 0, false
}

// VARCHAR group keys that are no small dictionary (Q18's c_name) come back as ROWS of their column
// (ph_plan_key_info: type PH_STR, the table and column): fetch those strings, one per group, and let the group's
// key value index them — the same decoding as gpuResidentPlanExecutor::Execute in plan_amd/csrc/host/executors.cpp.
func (e *gpuResidentPlanExecutor) decodeStringKeys(r *C.ph_agg_result) error {
	ng, nk := int(r.ngroups), maxInt(int(r.nkeys), 1)
	if ng == 0 {
		return nil
	}
	keys := unsafe.Slice((*int64)(unsafe.Pointer(r.keys)), ng*nk)
	for k := 0; k < e.nGroups; k++ {
		var kt, ks, kc C.int32_t
		var tab *C.ph_table
		if err := phErr(C.ph_plan_key_info(e.plan, C.int32_t(k), &kt, &ks, &tab, &kc)); err != nil {
			return err
		}
		if kt != C.PH_STR {
			continue
		}
		rows := make([]C.int64_t, ng)
		for g := 0; g < ng; g++ {
			rows[g] = C.int64_t(keys[g*nk+k])
		}
		off := make([]C.int32_t, ng+1)
		buf := make([]byte, 1<<20)
		if err := phErr(C.ph_table_strings(e.ctx, tab, kc, &rows[0], C.int64_t(ng), &off[0], (*C.char)(unsafe.Pointer(&buf[0])), C.int64_t(len(buf)))); err != nil {
			return err
		}
		dict := make([]string, ng)
		for g := 0; g < ng; g++ {
			dict[g] = string(buf[off[g]:off[g+1]])
			keys[g*nk+k] = int64(g)
		}
		e.keyDicts[k] = dict
	}
	return nil
}

func (b *planBuilder) setPreds(n *C.ph_plan_node, preds []C.ph_pred, bools []C.ph_bool) {
	if len(preds) > 0 {
		p := (*C.ph_pred)(b.arena.alloc(uintptr(len(preds)) * unsafe.Sizeof(preds[0])))
		copy(unsafe.Slice(p, len(preds)), preds)
		n.npreds, n.preds = C.int32_t(len(preds)), p
	}
	if len(bools) > 0 {
		p := (*C.ph_bool)(b.arena.alloc(uintptr(len(bools)) * unsafe.Sizeof(bools[0])))
		copy(unsafe.Slice(p, len(bools)), bools)
		n.nbools, n.bools = C.int32_t(len(bools)), p
	}
}

// ---------------------------------------------------------------------------------- the executor

type gpuResidentPlanExecutor struct {
	op       *PhysicalOperator
	ctx      *C.ph_ctx
	plan     *C.ph_plan
	result   *C.ph_agg_result
	next     int
	nGroups  int
	kinds    []C.int32_t
	argTypes []common.LType
	rowTypes []common.LType // [group columns | aggregate results]
	keyDicts [][]string
	havingEx *ExprExec
	outputEx *ExprExec
	sel      *chunk.SelectVector
}

// hasJoinBelow: Agg <- Scan alone is gpuScanAggExecutor's shape (the same fused kernels, one call less)
func hasJoinBelow(op *PhysicalOperator) bool {
	if op.Typ == POT_Join {
		return true
	}
	for _, c := range op.Children {
		if hasJoinBelow(c) {
			return true
		}
	}
	return false
}

// tryNewGpuResidentPlanExecutor is called by the POT_Agg arm of buildOperatorExec before tryNewGpuScanAggExecutor
// (INTEGRATION.md §2). parent: the operator above the Agg, when the caller knows it — an Order whose first key is one
// of the aggregates under a Limit announces the top-k preselection.
func tryNewGpuResidentPlanExecutor(op *PhysicalOperator, cfg *util.Config, txn *storage.Txn, children []OperatorExec, topK *planTopK) (OperatorExec, error) {
	if !gpuEnabled || op.Typ != POT_Agg || len(op.Children) != 1 || !hasJoinBelow(op.Children[0]) {
		return nil, errFallback
	}
	info, ok := op.Info.(*AggOpInfo)
	if !ok {
		return nil, errFallback
	}
	b := &planBuilder{cfg: cfg, txn: txn}
	defer b.arena.free()
	child, ok := b.lower(op.Children[0])
	if !ok {
		return nil, errFallback
	}
	e := &gpuResidentPlanExecutor{op: op}
	// group-by expressions and aggregates over the child's output columns
	groups := make([]C.ph_plan_expr, 0, len(info.GroupBys))
	for _, g := range info.GroupBys {
		if stripCast(g).Typ == ET_Const { // the constant key of an ungrouped aggregate (executor_aggr.go:37-48)
			continue
		}
		x, ok := b.lowerPlanExpr(g, 0)
		if !ok {
			return nil, errFallback
		}
		groups = append(groups, x)
		e.rowTypes = append(e.rowTypes, g.DataTyp)
		if x.kind == C.PH_PE_COL {
			e.keyDicts = append(e.keyDicts, b.meta[child].dicts[int(x.col)])
		} else {
			e.keyDicts = append(e.keyDicts, nil)
		}
	}
	e.nGroups = len(groups)
	aggs := make([]C.ph_plan_agg, 0, len(info.Aggs))
	for _, a := range info.Aggs {
		if a == nil || a.Typ != ET_Func || a.GetFuncInfo().FunImpl == nil {
			return nil, errFallback
		}
		kind, ok := aggKinds[strings.ToLower(a.FuncName())]
		if !ok {
			return nil, errFallback
		}
		if a.GetFuncInfo().FunImpl.IsDistinct() { // count(distinct x): PH_A_COUNT_DISTINCT (aggregate_exec.go:76-105, 201-304 inside the library)
			if kind != C.PH_A_COUNT || len(a.Children) != 1 {
				return nil, errFallback
			}
			kind = C.PH_A_COUNT_DISTINCT
		}
		var pa C.ph_plan_agg
		at := common.IntegerType()
		if kind == C.PH_A_COUNT && (len(a.Children) == 0 || stripCast(a.Children[0]).Typ == ET_Const) {
			pa.kind = C.PH_A_COUNT_STAR
		} else {
			if len(a.Children) != 1 {
				return nil, errFallback
			}
			x, ok := b.lowerPlanExpr(a.Children[0], 0)
			if !ok {
				return nil, errFallback
			}
			pa.kind, pa.arg = kind, x
			at = a.Children[0].DataTyp
		}
		aggs = append(aggs, pa)
		e.kinds = append(e.kinds, pa.kind)
		e.argTypes = append(e.argTypes, at)
		e.rowTypes = append(e.rowTypes, a.DataTyp)
	}
	var root C.ph_plan_node
	root.kind = C.PH_PN_AGG
	root.child[0], root.child[1] = C.int32_t(child), -1
	if len(groups) > 0 {
		gp := (*C.ph_plan_expr)(b.arena.alloc(uintptr(len(groups)) * unsafe.Sizeof(groups[0])))
		copy(unsafe.Slice(gp, len(groups)), groups)
		root.ngroups, root.groups = C.int32_t(len(groups)), gp
	}
	ap := (*C.ph_plan_agg)(b.arena.alloc(uintptr(len(aggs)) * unsafe.Sizeof(aggs[0])))
	copy(unsafe.Slice(ap, len(aggs)), aggs)
	root.naggs, root.aggs = C.int32_t(len(aggs)), ap
	b.add(root, planNodeMeta{})
	var err error
	if e.ctx, err = newGpuCtx(); err != nil {
		return nil, err
	}
	// the node array itself must be C memory too (it holds C pointers)
	np := (*C.ph_plan_node)(b.arena.alloc(uintptr(len(b.nodes)) * unsafe.Sizeof(b.nodes[0])))
	copy(unsafe.Slice(np, len(b.nodes)), b.nodes)
	if err = fallbackIf(C.ph_plan_create(e.ctx, np, C.int32_t(len(b.nodes)), &e.plan)); err != nil {
		C.ph_ctx_destroy(e.ctx)
		return nil, err
	}
	// HAVING (op.Filters of the aggregate) runs in the aggregate's output phase, before Order and Limit (executor_aggr.go:143-263):
	// with one, the top-k preselection is not announced — the k best groups could fail it while later ones pass.
	if topK != nil && topK.aggIndex >= 0 && len(op.Filters) == 0 {
		desc := C.int32_t(0)
		if topK.descending {
			desc = 1
		}
		if err = phErr(C.ph_plan_set_topk(e.plan, C.int32_t(topK.aggIndex), desc, C.int64_t(topK.k))); err != nil {
			e.Close()
			return nil, err
		}
	}
	_ = children // the CPU executors built for the subtree are not pulled; buildOperatorExec closes them with the tree
	return e, nil
}

// planTopK: ORDER BY <aggregate aggIndex> [DESC] ... LIMIT k above the aggregate (executor_order.go:56-138 +
// executor_limit.go:105-238): only the groups that can reach the first k rows come back; Order and Limit still run.
type planTopK struct {
	aggIndex   int
	descending bool
	k          int64
}

// HAVING conjuncts of the form `aggregate OP numeric literal` go down to the plan (ph_plan_set_having: evaluated on the device when the
// groups are fetched, with selectOperation's own rules — '>' is the one comparison DECIMAL and HUGEINT have, a DECIMAL against a FLOAT
// literal compares in float32); anything else — a conjunct over a group key, an OR, an AVG — keeps the reference's ExprExec below.
// In the aggregate's output phase a column reference addresses [group columns (table -1) | - | aggregate results (table -3)]
// (executor_aggr.go:143-263): the result column of ph_pred counts the group keys first.
func (e *gpuResidentPlanExecutor) pushHaving() bool {
	var ptrs []unsafe.Pointer
	preds := make([]C.ph_pred, 0, len(e.op.Filters))
	for _, f := range e.op.Filters {
		if f == nil || f.Typ != ET_Func || len(f.Children) != 2 {
			return false
		}
		op, isCmp := cmpOps[f.FuncName()]
		tab, col, isCol := colRefOf(stripCast(f.Children[0]))
		if !isCmp || !isCol || tab != -3 {
			return false
		}
		k, ok := lowerConst(f.Children[1], &ptrs)
		if !ok || len(ptrs) > 0 || k._type == C.PH_STR || k._type == C.PH_DATE {
			for _, q := range ptrs {
				C.free(q)
			}
			return false
		}
		var p C.ph_pred
		p.col, p.op, p.k = C.int32_t(e.nGroups+col), op, k
		preds = append(preds, p)
	}
	return len(preds) > 0 && C.ph_plan_set_having(e.plan, C.int32_t(len(preds)), &preds[0]) == C.PH_OK
}

func (e *gpuResidentPlanExecutor) Init() error {
	if len(e.op.Filters) > 0 && !e.pushHaving() {
		e.havingEx = NewExprExec(e.op.Filters...)
		e.sel = chunk.NewSelectVector(util.DefaultVectorSize)
	}
	e.outputEx = NewExprExec(e.op.Outputs...)
	return nil
}

func (e *gpuResidentPlanExecutor) Execute(input, output *chunk.Chunk) (OperatorResult, error) {
	ensureOutputChunk(e.op, output)
	if e.result == nil { // pipeline breaker, like HAS_INIT (executor_aggr.go:110-142); a broken statistic reruns inside fetch
		if err := phErr(C.ph_plan_run(e.plan)); err != nil {
			return InvalidOpResult, err
		}
		if err := phErr(C.ph_plan_fetch(e.plan, &e.result)); err != nil {
			return InvalidOpResult, err
		}
		// a sum beyond int64 makes the device hand every group back unfiltered (ph_plan_having_applied = 0): the HAVING then runs
		// here through the reference's own ExprExec, like one that never went down
		if e.havingEx == nil && len(e.op.Filters) > 0 && C.ph_plan_having_applied(e.plan) == 0 {
			e.havingEx = NewExprExec(e.op.Filters...)
			e.sel = chunk.NewSelectVector(util.DefaultVectorSize)
		}
		if err := e.decodeStringKeys(e.result); err != nil {
			return InvalidOpResult, err
		}
	}
	r := e.result
	ng, nk, na := int(r.ngroups), maxInt(int(r.nkeys), 1), int(r.naggs)
	for e.next < ng {
		n := ng - e.next
		if n > util.DefaultVectorSize {
			n = util.DefaultVectorSize
		}
		rows := &chunk.Chunk{}
		rows.Init(e.rowTypes, util.DefaultVectorSize)
		keys := unsafe.Slice((*int64)(unsafe.Pointer(r.keys)), ng*nk)
		lo := unsafe.Slice((*uint64)(unsafe.Pointer(r.sum_lo)), ng*na)
		hi := unsafe.Slice((*int64)(unsafe.Pointer(r.sum_hi)), ng*na)
		cnt := unsafe.Slice((*uint64)(unsafe.Pointer(r.count)), ng*na)
		scales := make([]int, na)
		for a := 0; a < na; a++ {
			scales[a] = int(*(*C.int32_t)(unsafe.Add(unsafe.Pointer(r.scale), 4*a)))
		}
		var keyNull []uint8 // the NULL group of a NULL-able key (Q13's c_count: a count of 0 finalises to NULL)
		if r.key_null != nil {
			keyNull = unsafe.Slice((*uint8)(unsafe.Pointer(r.key_null)), ng*nk)
		}
		if err := emitGroupRows(rows, e.rowTypes[:e.nGroups], e.keyDicts, e.kinds, e.argTypes, scales, nk, na, keys, keyNull, lo, hi, cnt, e.next, n); err != nil {
			return InvalidOpResult, err
		}
		e.next += n
		// output phase through the reference's own ExprExec, as gpuScanAggExecutor: HAVING over the aggregate results,
		// then the output expressions — Q14's FLOAT arithmetic over the two sums runs here, on one row
		groupChunk, aggrChunk := &chunk.Chunk{}, &chunk.Chunk{}
		groupChunk.Init(e.rowTypes[:e.nGroups], util.DefaultVectorSize)
		aggrChunk.Init(e.rowTypes[e.nGroups:], util.DefaultVectorSize)
		for i := 0; i < e.nGroups; i++ {
			groupChunk.Data[i].Reference(rows.Data[i])
		}
		for i := 0; i < na; i++ {
			aggrChunk.Data[i].Reference(rows.Data[e.nGroups+i])
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

// Explain: the library's account of the forms the last run chose (one line per operator)
func (e *gpuResidentPlanExecutor) Explain() string {
	if e.plan == nil {
		return ""
	}
	return C.GoString(C.ph_plan_explain(e.plan))
}

func (e *gpuResidentPlanExecutor) Close() error {
	if e.result != nil {
		C.ph_agg_result_free(e.result)
	}
	if e.plan != nil {
		C.ph_plan_free(e.plan)
	}
	if e.ctx != nil {
		C.ph_ctx_destroy(e.ctx)
	}
	e.result, e.plan, e.ctx = nil, nil, nil
	return nil
}
