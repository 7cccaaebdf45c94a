// Expression: the subset of arrow::compute::Expression (compute/exec/expression.h:40-252, expression.cc) that the dataset
// scanner's per-batch step needs (dataset/scanner.cc:456-500): build a tree of field references, literals and function
// calls, Bind it to a schema (types resolved through DispatchBest), ExecuteScalarExpression it against a RecordBatch whose
// columns live in HBM, and FilterAndProjectRecordBatch = mask -> Filter -> projection.  The planner half of the reference
// (Canonicalize, FoldConstants, SimplifyWithGuarantee, Serialize) works on the tree only and stays with the CPU caller.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "amdarrow_compute.h"

namespace amdarrow {
namespace compute {

struct MakeStructOptions : public FunctionOptions {  // api_scalar.h:236-250
  explicit MakeStructOptions(std::vector<std::string> n = {}) : field_names(std::move(n)) {}
  std::vector<std::string> field_names;
};

class Expression {
 public:
  struct Call {
    std::string function_name;
    std::vector<Expression> arguments;
    std::shared_ptr<FunctionOptions> options;
  };
  Expression() = default;
  bool is_valid() const { return impl_ != nullptr; }
  const Datum* literal() const;            // expression.h:78
  const std::string* field_ref() const;    // expression.h:81 (by name)
  const Call* call() const;                // expression.h:75
  // expression.h:52-62 / expression.cc:485-511: field references resolved against the schema (an absent field is a null
  // scalar), calls resolved through the registry's DispatchBest; the result carries its type.
  Result<Expression> Bind(const Schema& in_schema, ExecContext* ctx) const;
  bool IsBound() const;                    // expression.cc:130-147
  std::shared_ptr<DataType> type() const;  // of a bound expression
  std::string ToString() const;

 private:
  struct Impl;
  std::shared_ptr<const Impl> impl_;
  friend Expression literal(Datum lit);
  friend Expression field_ref(std::string name);
  friend Expression call(std::string function, std::vector<Expression> arguments, std::shared_ptr<FunctionOptions> options);
};

Expression literal(Datum lit);
template <typename T>
Expression literal(std::shared_ptr<DataType> type, T value) { return literal(Datum(Scalar::Make<T>(std::move(type), value))); }
Expression field_ref(std::string name);
Expression call(std::string function, std::vector<Expression> arguments, std::shared_ptr<FunctionOptions> options = nullptr);

std::vector<std::string> FieldsInExpression(const Expression&);  // expression.cc:601-614
bool ExpressionHasFieldRefs(const Expression&);                  // :616-625

// expression.cc:513-566.  input: a RecordBatch (Datum::RECORD_BATCH).  "make_struct" calls (project) yield a struct array.
Result<Datum> ExecuteScalarExpression(const Expression& expr, const Datum& input, ExecContext* ctx);

// expression.h:226-249
Expression project(std::vector<Expression> values, std::vector<std::string> names);
Expression equal(Expression lhs, Expression rhs);
Expression not_equal(Expression lhs, Expression rhs);
Expression less(Expression lhs, Expression rhs);
Expression less_equal(Expression lhs, Expression rhs);
Expression greater(Expression lhs, Expression rhs);
Expression greater_equal(Expression lhs, Expression rhs);
Expression is_null(Expression lhs);
Expression is_valid(Expression lhs);
Expression and_(Expression lhs, Expression rhs);
Expression and_(const std::vector<Expression>&);
Expression or_(Expression lhs, Expression rhs);
Expression or_(const std::vector<Expression>&);
Expression not_(Expression operand);

// The scanner's per-batch step (dataset/scanner.cc:456-500 DoFilterAndProjectRecordBatchAsync): evaluate the bound filter,
// keep the rows it selects (a scalar mask keeps all or none), evaluate the bound projection on what is left.
Result<std::shared_ptr<RecordBatch>> FilterAndProjectRecordBatch(const std::shared_ptr<RecordBatch>& batch, const Expression& filter,
                                                                 const Expression& projection, ExecContext* ctx);

}  // namespace compute
}  // namespace amdarrow
