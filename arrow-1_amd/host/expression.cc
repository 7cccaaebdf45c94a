// See expression.h.  Reference: compute/exec/expression.cc (Bind :485-511, ExecuteScalarExpression :513-566,
// FieldsInExpression :601-625, the factory helpers :1200-1290) and dataset/scanner.cc:456-500.
#include "expression.h"

#include <utility>

namespace amdarrow {
namespace compute {

struct Expression::Impl {
  enum Kind { LITERAL, FIELD, CALL } kind = LITERAL;
  Datum lit;
  std::string name;
  Call call;
  // set by Bind
  bool bound = false;
  int field_index = -1;  // FIELD: column of the schema, -1 = absent (evaluates to a null scalar)
  ValueDescr descr;
};

const Datum* Expression::literal() const { return impl_ && impl_->kind == Impl::LITERAL ? &impl_->lit : nullptr; }
const std::string* Expression::field_ref() const { return impl_ && impl_->kind == Impl::FIELD ? &impl_->name : nullptr; }
const Expression::Call* Expression::call() const { return impl_ && impl_->kind == Impl::CALL ? &impl_->call : nullptr; }

Expression literal(Datum lit) {
  auto impl = std::make_shared<Expression::Impl>();
  impl->kind = Expression::Impl::LITERAL;
  impl->lit = std::move(lit);
  impl->bound = true;  // a literal is bound by construction (expression.cc:492)
  impl->descr = impl->lit.descr();
  Expression e;
  e.impl_ = std::move(impl);
  return e;
}
Expression field_ref(std::string name) {
  auto impl = std::make_shared<Expression::Impl>();
  impl->kind = Expression::Impl::FIELD;
  impl->name = std::move(name);
  Expression e;
  e.impl_ = std::move(impl);
  return e;
}
Expression call(std::string function, std::vector<Expression> arguments, std::shared_ptr<FunctionOptions> options) {
  auto impl = std::make_shared<Expression::Impl>();
  impl->kind = Expression::Impl::CALL;
  impl->call.function_name = std::move(function);
  impl->call.arguments = std::move(arguments);
  impl->call.options = std::move(options);
  Expression e;
  e.impl_ = std::move(impl);
  return e;
}

bool Expression::IsBound() const {
  if (!impl_ || !impl_->bound) return false;
  if (impl_->kind == Impl::CALL)
    for (const auto& a : impl_->call.arguments)
      if (!a.IsBound()) return false;
  return true;
}
std::shared_ptr<DataType> Expression::type() const { return impl_ && impl_->bound ? impl_->descr.type : nullptr; }

std::string Expression::ToString() const {
  if (!impl_) return "<invalid>";
  if (impl_->kind == Impl::LITERAL) return impl_->lit.is_scalar() ? "literal(" + impl_->lit.type()->ToString() + ")" : "literal(<array>)";
  if (impl_->kind == Impl::FIELD) return impl_->name;
  std::string s = impl_->call.function_name + "(";
  for (size_t i = 0; i < impl_->call.arguments.size(); ++i) s += (i ? ", " : "") + impl_->call.arguments[i].ToString();
  return s + ")";
}

Result<Expression> Expression::Bind(const Schema& in_schema, ExecContext* ctx) const {
  if (!impl_) return Status::Invalid("Cannot Bind an invalid expression");
  if (impl_->kind == Impl::LITERAL) return *this;
  auto bound = std::make_shared<Impl>(*impl_);
  bound->bound = true;
  if (impl_->kind == Impl::FIELD) {
    bound->field_index = in_schema.GetFieldIndex(impl_->name);
    bound->descr = bound->field_index >= 0 ? ValueDescr::Array(in_schema.field(bound->field_index).type) : ValueDescr::Scalar(null());
  } else {
    std::vector<ValueDescr> descrs;
    for (auto& a : bound->call.arguments) {
      AMD_ASSIGN_OR_RAISE(a, a.Bind(in_schema, ctx));
      descrs.push_back(a.impl_->descr);
    }
    bool any_array = false;
    for (const auto& d : descrs) any_array |= d.shape == ValueDescr::ARRAY;
    if (bound->call.function_name == "make_struct") {
      bound->descr = any_array ? ValueDescr::Array(struct_()) : ValueDescr::Scalar(struct_());
    } else {
      AMD_ASSIGN_OR_RAISE(auto fn, ctx->func_registry()->GetFunction(bound->call.function_name));
      if (fn->kind() != Function::SCALAR) return Status::Invalid("Expression call to the non-scalar function " + fn->name());
      const auto* sf = static_cast<const ScalarFunction*>(fn.get());
      AMD_ASSIGN_OR_RAISE(const ScalarKernel* kernel, sf->DispatchBest(&descrs));  // implicit casts happen at Execute
      AMD_ASSIGN_OR_RAISE(auto out_type, kernel->signature.out_type(descrs));
      bound->descr = any_array ? ValueDescr::Array(out_type) : ValueDescr::Scalar(out_type);
    }
  }
  Expression e;
  e.impl_ = std::move(bound);
  return e;
}

std::vector<std::string> FieldsInExpression(const Expression& expr) {
  if (expr.literal()) return {};
  if (auto ref = expr.field_ref()) return {*ref};
  std::vector<std::string> fields;
  if (auto c = expr.call())
    for (const auto& a : c->arguments) {
      auto sub = FieldsInExpression(a);
      fields.insert(fields.end(), sub.begin(), sub.end());
    }
  return fields;
}
bool ExpressionHasFieldRefs(const Expression& expr) { return !FieldsInExpression(expr).empty(); }

Result<Datum> ExecuteScalarExpression(const Expression& expr, const Datum& input, ExecContext* ctx) {
  if (!expr.IsBound()) return Status::Invalid("Cannot Execute unbound expression.");
  if (input.kind() != Datum::RECORD_BATCH) return Status::NotImplemented("ExecuteScalarExpression over a record batch only");
  if (auto lit = expr.literal()) return *lit;
  const auto& batch = input.record_batch();
  if (auto ref = expr.field_ref()) {
    // by name again: the batch at hand may lay its columns out differently from the schema of Bind (GetDatumField)
    auto col = batch->GetColumnByName(*ref);
    if (!col) return Datum(Scalar::MakeNull(null()));
    return Datum(col);
  }
  const auto* c = expr.call();
  std::vector<Datum> args;
  for (const auto& a : c->arguments) {
    AMD_ASSIGN_OR_RAISE(Datum v, ExecuteScalarExpression(a, input, ctx));
    args.push_back(std::move(v));
  }
  if (c->function_name == "make_struct") {  // scalar_nested.cc MakeStruct: the arguments side by side under the given names
    const auto* o = static_cast<const MakeStructOptions*>(c->options.get());
    if (o == nullptr || o->field_names.size() != args.size()) return Status::Invalid("make_struct: one field name per value");
    auto out = ArrayData::Make(struct_(), batch->num_rows(), {nullptr, nullptr}, 0);
    for (size_t i = 0; i < args.size(); ++i) {
      if (!args[i].is_array()) return Status::NotImplemented("make_struct of a scalar value (broadcast is left to the caller)");
      out->child_data.push_back(args[i].array());
    }
    out->child_names = o->field_names;
    return Datum(out);
  }
  return CallFunction(c->function_name, args, c->options.get(), ctx);
}

Expression project(std::vector<Expression> values, std::vector<std::string> names) {
  return call("make_struct", std::move(values), std::make_shared<MakeStructOptions>(std::move(names)));
}
Expression equal(Expression lhs, Expression rhs) { return call("equal", {std::move(lhs), std::move(rhs)}); }
Expression not_equal(Expression lhs, Expression rhs) { return call("not_equal", {std::move(lhs), std::move(rhs)}); }
Expression less(Expression lhs, Expression rhs) { return call("less", {std::move(lhs), std::move(rhs)}); }
Expression less_equal(Expression lhs, Expression rhs) { return call("less_equal", {std::move(lhs), std::move(rhs)}); }
Expression greater(Expression lhs, Expression rhs) { return call("greater", {std::move(lhs), std::move(rhs)}); }
Expression greater_equal(Expression lhs, Expression rhs) { return call("greater_equal", {std::move(lhs), std::move(rhs)}); }
Expression is_null(Expression lhs) { return call("is_null", {std::move(lhs)}); }
Expression is_valid(Expression lhs) { return call("is_valid", {std::move(lhs)}); }
Expression and_(Expression lhs, Expression rhs) { return call("and_kleene", {std::move(lhs), std::move(rhs)}); }
Expression or_(Expression lhs, Expression rhs) { return call("or_kleene", {std::move(lhs), std::move(rhs)}); }
Expression not_(Expression operand) { return call("invert", {std::move(operand)}); }
static Expression FoldLeft(const std::vector<Expression>& operands, bool neutral, Expression (*op)(Expression, Expression)) {
  if (operands.empty()) return literal(Datum(Scalar::Make<uint8_t>(boolean(), neutral ? 1 : 0)));  // expression.cc:1262-1282
  Expression folded = operands[0];
  for (size_t i = 1; i < operands.size(); ++i) folded = op(std::move(folded), operands[i]);
  return folded;
}
Expression and_(const std::vector<Expression>& operands) { return FoldLeft(operands, true, and_); }
Expression or_(const std::vector<Expression>& operands) { return FoldLeft(operands, false, or_); }

Result<std::shared_ptr<RecordBatch>> FilterAndProjectRecordBatch(const std::shared_ptr<RecordBatch>& batch, const Expression& filter,
                                                                 const Expression& projection, ExecContext* ctx) {
  AMD_ASSIGN_OR_RAISE(Datum mask, ExecuteScalarExpression(filter, Datum(batch), ctx));
  std::shared_ptr<RecordBatch> filtered;
  if (mask.is_scalar()) {
    const Scalar& m = *mask.scalar();
    if (m.is_valid && m.type->id() == Type::BOOL && (m.bits & 1)) {
      filtered = batch;  // the filter matches the entire batch
    } else {               // ... or nothing: Slice(0, 0)
      std::vector<std::shared_ptr<ArrayData>> none;
      for (int i = 0; i < batch->num_columns(); ++i) none.push_back(batch->column(i)->Slice(0, 0));
      filtered = RecordBatch::Make(batch->schema(), 0, std::move(none));
    }
  } else {
    AMD_ASSIGN_OR_RAISE(Datum kept, Filter(Datum(batch), mask, FilterOptions::Defaults(), ctx));
    filtered = kept.record_batch();
  }
  AMD_ASSIGN_OR_RAISE(Datum projected, ExecuteScalarExpression(projection, Datum(filtered), ctx));
  if (!projected.is_array() || projected.type()->id() != Type::STRUCT) return Status::Invalid("the projection must be a make_struct call");
  const ArrayData& s = *projected.array();
  std::vector<Field> fields;
  for (size_t i = 0; i < s.child_data.size(); ++i) fields.push_back(field(s.child_names[i], s.child_data[i]->type));
  return RecordBatch::Make(schema(std::move(fields)), filtered->num_rows(), s.child_data);
}

}  // namespace compute
}  // namespace amdarrow
