//! `Arc<dyn PhysicalExpr>` -> the flat `dfgpu_expr_node[]` IR (+ the pool of string literals).  `None` = not offloadable: the
//! rule keeps the CPU operator.  Python twin: datafusion_amd/expr.py `lower`.
use crate::sys::*;
use arrow::datatypes::{DataType, Schema};
use datafusion::common::ScalarValue;
use datafusion::logical_expr::Operator;
use datafusion::physical_expr::expressions::{BinaryExpr, CaseExpr, CastExpr, Column, InListExpr, IsNotNullExpr, IsNullExpr, LikeExpr, Literal, NotExpr};
use datafusion::physical_expr::{PhysicalExpr, ScalarFunctionExpr};
use std::sync::Arc;

#[derive(Default)]
pub struct Lowered {
    pub nodes: Vec<dfgpu_expr_node>,
    pub pool: Vec<u8>,
}
impl std::fmt::Debug for Lowered {
    fn fmt(&self, f: &mut std::fmt::Formatter) -> std::fmt::Result {
        write!(f, "Lowered({} nodes: {:?})", self.nodes.len(), self.nodes.iter().map(|n| n.op).collect::<Vec<_>>())
    }
}
impl Lowered {
    pub fn as_c(&self) -> dfgpu_expr {
        // the root is the last node (post-order flattening below); the pool pointer is only read when a node refers into it
        dfgpu_expr { nodes: self.nodes.as_ptr(), n_nodes: self.nodes.len() as i32, root: self.nodes.len() as i32 - 1, string_pool: self.pool.as_ptr() as *const _ }
    }
}

pub fn field_of(t: &DataType) -> Option<dfgpu_field> {
    let (type_, precision, scale) = match t {
        DataType::Int32 => (DFGPU_INT32, 0, 0),
        DataType::Int64 => (DFGPU_INT64, 0, 0),
        DataType::Decimal128(p, s) => (DFGPU_DECIMAL128, *p as i32, *s as i32),
        DataType::Float64 => (DFGPU_FLOAT64, 0, 0),
        DataType::UInt8 => (DFGPU_UINT8, 0, 0),
        DataType::UInt32 => (DFGPU_UINT32, 0, 0),
        DataType::UInt64 => (DFGPU_UINT64, 0, 0),
        DataType::Date32 => (DFGPU_DATE32, 0, 0),
        DataType::Boolean => (DFGPU_BOOL, 0, 0),
        DataType::Utf8 | DataType::LargeUtf8 | DataType::Utf8View => (DFGPU_UTF8, 0, 0),
        DataType::Dictionary(k, v) if matches!(**v, DataType::Utf8 | DataType::LargeUtf8) => return field_of(k), // indices on the device
        _ => return None,
    };
    Some(dfgpu_field { type_, precision, scale, nullable: 1 })
}

fn node(op: dfgpu_expr_op) -> dfgpu_expr_node {
    dfgpu_expr_node { op, column: -1, left: -1, right: -1, field: dfgpu_field { type_: 0, precision: 0, scale: 0, nullable: 1 }, is_null: 0, _pad: 0, lit_lo: 0, lit_hi: 0 }
}

/// post-order flattening; the root is the last node
pub fn lower(e: &Arc<dyn PhysicalExpr>, schema: &Schema, out: &mut Lowered) -> Option<i32> {
    let any = e.as_ref(); // `dyn PhysicalExpr`: downcast_ref is the trait object's own helper in 55 (physical_expr.rs:814)
    let mut n;
    if let Some(c) = any.downcast_ref::<Column>() {
        field_of(schema.field(c.index()).data_type())?;
        n = node(DFGPU_EXPR_COLUMN);
        n.column = c.index() as i32;
    } else if let Some(l) = any.downcast_ref::<Literal>() {
        n = node(DFGPU_EXPR_LITERAL);
        n.field = field_of(&l.value().data_type())?;
        match l.value() {
            v if v.is_null() => n.is_null = 1,
            ScalarValue::Utf8(Some(s)) | ScalarValue::LargeUtf8(Some(s)) | ScalarValue::Utf8View(Some(s)) => {
                n.lit_lo = out.pool.len() as u64; // (offset, length) into dfgpu_expr.string_pool
                n.lit_hi = s.len() as u64;
                out.pool.extend_from_slice(s.as_bytes());
            }
            ScalarValue::Float64(Some(f)) => n.lit_lo = f.to_bits(),
            ScalarValue::Decimal128(Some(v), _, _) => (n.lit_lo, n.lit_hi) = (*v as u128 as u64, (*v as u128 >> 64) as u64),
            v => {
                let x = scalar_as_i128(v)? as u128; // integers / Date32 / Boolean sign-extended to 128 bits
                (n.lit_lo, n.lit_hi) = (x as u64, (x >> 64) as u64)
            }
        }
    } else if let Some(c) = any.downcast_ref::<CastExpr>() {
        n = node(DFGPU_EXPR_CAST);
        n.left = lower(c.expr(), schema, out)?;
        n.field = field_of(c.cast_type())?;
    } else if let Some(b) = any.downcast_ref::<BinaryExpr>() {
        n = node(match b.op() {
            Operator::Plus => DFGPU_EXPR_ADD, Operator::Minus => DFGPU_EXPR_SUB, Operator::Multiply => DFGPU_EXPR_MUL,
            Operator::Divide => DFGPU_EXPR_DIV, Operator::Modulo => DFGPU_EXPR_MOD,
            Operator::Eq => DFGPU_EXPR_EQ, Operator::NotEq => DFGPU_EXPR_NE, Operator::Lt => DFGPU_EXPR_LT, Operator::LtEq => DFGPU_EXPR_LE,
            Operator::Gt => DFGPU_EXPR_GT, Operator::GtEq => DFGPU_EXPR_GE, Operator::And => DFGPU_EXPR_AND, Operator::Or => DFGPU_EXPR_OR,
            _ => return None,
        });
        n.left = lower(b.left(), schema, out)?;
        n.right = lower(b.right(), schema, out)?;
    } else if let Some(x) = any.downcast_ref::<NotExpr>() {
        n = node(DFGPU_EXPR_NOT);
        n.left = lower(x.arg(), schema, out)?;
    } else if let Some(x) = any.downcast_ref::<IsNullExpr>() {
        n = node(DFGPU_EXPR_IS_NULL);
        n.left = lower(x.arg(), schema, out)?;
    } else if let Some(x) = any.downcast_ref::<IsNotNullExpr>() {
        n = node(DFGPU_EXPR_IS_NOT_NULL);
        n.left = lower(x.arg(), schema, out)?;
    } else if let Some(l) = any.downcast_ref::<LikeExpr>() {
        // on a Utf8 column: matched on the bytes in HBM; on a dictionary column the caller binds the pattern to index ranges
        // first (dfgpu_table_dictionary_like), exactly as datafusion_amd/expr.py LikeExpr.bound does
        n = node(if l.case_insensitive() { DFGPU_EXPR_ILIKE } else { DFGPU_EXPR_LIKE });
        n.left = lower(l.expr(), schema, out)?;
        n.right = lower(l.pattern(), schema, out)?;
        if l.negated() {
            out.nodes.push(n);
            n = node(DFGPU_EXPR_NOT);
            n.left = out.nodes.len() as i32 - 1;
        }
    } else if let Some(f) = any.downcast_ref::<ScalarFunctionExpr>() {
        if f.name() == "substr" {
            // substr(string column, Int64 start [, Int64 count]) with literal positions (functions/src/unicode/substr.rs)
            let int = |e: &Arc<dyn PhysicalExpr>| -> Option<i64> {
                match e.downcast_ref::<Literal>()?.value() { ScalarValue::Int64(Some(v)) => Some(*v), _ => None }
            };
            if f.args().len() < 2 || f.args().len() > 3 { return None; }
            n = node(DFGPU_EXPR_SUBSTR);
            n.left = lower(&f.args()[0], schema, out)?;
            n.column = i32::try_from(int(&f.args()[1])?).ok()?;
            match f.args().get(2) {
                Some(c) => { let v = int(c)?; n.lit_lo = v as u64; n.lit_hi = if v < 0 { u64::MAX } else { 0 }; }
                None => n.is_null = 1,
            }
            out.nodes.push(n);
            return Some(out.nodes.len() as i32 - 1);
        }
        // date_part('year' | 'month' | 'day', Date32)
        if f.name() != "date_part" { return None; }
        let part = f.args()[0].downcast_ref::<Literal>()?.value().to_string().to_lowercase();
        n = node(DFGPU_EXPR_DATE_PART);
        n.column = match part.as_str() { "year" => DFGPU_DATE_PART_YEAR, "month" => DFGPU_DATE_PART_MONTH, "day" => DFGPU_DATE_PART_DAY, _ => return None };
        n.left = lower(&f.args()[1], schema, out)?;
    } else if let Some(c) = any.downcast_ref::<CaseExpr>() {
        // no base expression: one DFGPU_EXPR_CASE node per WHEN, later branches nested in ELSE
        if c.expr().is_some() { return None; }
        let mut tail = match c.else_expr() { Some(e) => lower(e, schema, out)?, None => -1 };
        for (w, t) in c.when_then_expr().iter().rev() {
            let mut m = node(DFGPU_EXPR_CASE);
            m.column = lower(w, schema, out)?;
            m.left = lower(t, schema, out)?;
            m.right = tail;
            out.nodes.push(m);
            tail = out.nodes.len() as i32 - 1;
        }
        return Some(tail);
    } else if let Some(l) = any.downcast_ref::<InListExpr>() {
        // short literal lists: the Kleene OR of equalities (NOT of it when negated) — SQL's definition, the reference's NULL rules
        if l.list().len() > 32 { return None; }
        let x = l.expr();
        let mut acc = -1;
        for v in l.list() {
            let mut eq = node(DFGPU_EXPR_EQ);
            eq.left = lower(x, schema, out)?;
            eq.right = lower(v, schema, out)?;
            out.nodes.push(eq);
            let this = out.nodes.len() as i32 - 1;
            if acc < 0 { acc = this; continue; }
            let mut or = node(DFGPU_EXPR_OR);
            (or.left, or.right) = (acc, this);
            out.nodes.push(or);
            acc = out.nodes.len() as i32 - 1;
        }
        if !l.negated() { return Some(acc); }
        n = node(DFGPU_EXPR_NOT);
        n.left = acc;
    } else {
        return None; // scalar UDFs, substr, regexp, nested types ...: the CPU operator is kept
    }
    out.nodes.push(n);
    Some(out.nodes.len() as i32 - 1)
}

fn scalar_as_i128(v: &ScalarValue) -> Option<i128> {
    Some(match v {
        ScalarValue::Int32(Some(x)) | ScalarValue::Date32(Some(x)) => *x as i128,
        ScalarValue::Int64(Some(x)) => *x as i128,
        ScalarValue::UInt8(Some(x)) => *x as i128,
        ScalarValue::UInt32(Some(x)) => *x as i128,
        ScalarValue::UInt64(Some(x)) => *x as i128,
        ScalarValue::Boolean(Some(x)) => *x as i128,
        _ => return None,
    })
}
