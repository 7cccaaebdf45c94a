# Builds the REFERENCE hot path (Arrow 5.0.0-SNAPSHOT compute kernels) straight from
# the sources where they lie under $(REF), with g++ only -- no cmake, no generated
# headers, no third-party libraries.  Output goes to oracle/_ref/ (git-ignored).
#
# TEST INFRASTRUCTURE ONLY.  Nothing under arrow-1_amd/ may link or load this.
#
# What is deliberately NOT compiled (and why the hot path does not need it):
#   util/basic_decimal.cc util/decimal.cc util/formatting.cc config.cc
#       -> they include arrow/util/config.h, which only cmake generates; the hot path
#          (filter/take/sort_indices/hash_aggregate/compare/arithmetic on primitive
#          types) never calls decimal or number-formatting code.
#   ipc/ csv/ json/ filesystem/ c/ tensor/ compute/exec/ and most of io/
#       -> IO, formats and the ExecPlan/Expression layer sit outside the path
#          (io/interfaces.cc + io/memory.cc are compiled only because
#          util/io_util.cc -- GetEnvVar for the memory pool -- needs their vtables).
# The shared object therefore carries unresolved symbols for those families
# (decimal, float formatting, the string/nested/temporal kernels' Register* hooks).
# Every object is passed through `objcopy --weaken`, which turns its undefined
# references into WEAK undefined ones, so the library loads even under RTLD_NOW
# (python ctypes always adds RTLD_NOW) with those symbols bound to null.  The
# driver (ref_driver.cc) never reaches them: it builds its own FunctionRegistry with FunctionRegistry::Make()
# and calls only the Register* entry points of the hot-path kernel files
# (compute/registry_internal.h), exactly the extension mechanism the reference's
# own tests use (compute/exec_test.cc:651-717).
#
# usage: make -f oracle/ref_build.mk -j8        (from the repo root)

REF    ?= /root/reference/cpp/src
OUT    ?= oracle/_ref
OBJ    := $(OUT)/obj
CXX    ?= g++
CXXFLAGS := -std=c++11 -O2 -fPIC -w -DNDEBUG -DARROW_EXPORTING -include limits -include cstdint -I$(REF)

A := $(REF)/arrow

SRCS := \
  $(A)/array/array_base.cc $(A)/array/array_binary.cc $(A)/array/array_decimal.cc \
  $(A)/array/array_dict.cc $(A)/array/array_nested.cc $(A)/array/array_primitive.cc \
  $(A)/array/builder_adaptive.cc $(A)/array/builder_base.cc $(A)/array/builder_binary.cc \
  $(A)/array/builder_decimal.cc $(A)/array/builder_dict.cc $(A)/array/builder_nested.cc \
  $(A)/array/builder_primitive.cc $(A)/array/builder_union.cc $(A)/array/concatenate.cc \
  $(A)/array/data.cc $(A)/array/diff.cc $(A)/array/util.cc $(A)/array/validate.cc \
  $(A)/buffer.cc $(A)/builder.cc $(A)/chunked_array.cc $(A)/compare.cc $(A)/datum.cc \
  $(A)/device.cc $(A)/extension_type.cc $(A)/memory_pool.cc $(A)/pretty_print.cc \
  $(A)/record_batch.cc $(A)/result.cc $(A)/scalar.cc $(A)/status.cc $(A)/table.cc \
  $(A)/type.cc $(A)/visitor.cc \
  $(A)/util/bit_block_counter.cc $(A)/util/bit_run_reader.cc $(A)/util/bit_util.cc \
  $(A)/util/bitmap.cc $(A)/util/bitmap_builders.cc $(A)/util/bitmap_ops.cc \
  $(A)/util/cancel.cc $(A)/util/cpu_info.cc $(A)/util/future.cc $(A)/util/int_util.cc \
  $(A)/util/io_util.cc $(A)/util/key_value_metadata.cc $(A)/util/logging.cc \
  $(A)/util/memory.cc $(A)/util/mutex.cc $(A)/util/string.cc $(A)/util/string_builder.cc \
  $(A)/util/task_group.cc $(A)/util/tdigest.cc $(A)/util/thread_pool.cc $(A)/util/time.cc \
  $(A)/util/utf8.cc $(A)/util/value_parsing.cc \
  $(A)/io/interfaces.cc $(A)/io/memory.cc \
  $(A)/vendored/double-conversion/bignum.cc $(A)/vendored/double-conversion/double-conversion.cc \
  $(A)/vendored/double-conversion/bignum-dtoa.cc $(A)/vendored/double-conversion/fast-dtoa.cc \
  $(A)/vendored/double-conversion/cached-powers.cc $(A)/vendored/double-conversion/fixed-dtoa.cc \
  $(A)/vendored/double-conversion/diy-fp.cc $(A)/vendored/double-conversion/strtod.cc \
  $(A)/compute/api_aggregate.cc $(A)/compute/api_scalar.cc $(A)/compute/api_vector.cc \
  $(A)/compute/cast.cc $(A)/compute/exec.cc $(A)/compute/function.cc $(A)/compute/kernel.cc \
  $(A)/compute/registry.cc \
  $(A)/compute/kernels/aggregate_basic.cc $(A)/compute/kernels/codegen_internal.cc \
  $(A)/compute/kernels/hash_aggregate.cc $(A)/compute/kernels/scalar_arithmetic.cc \
  $(A)/compute/kernels/scalar_boolean.cc $(A)/compute/kernels/scalar_compare.cc \
  $(A)/compute/kernels/scalar_validity.cc $(A)/compute/kernels/scalar_fill_null.cc \
  $(A)/compute/kernels/util_internal.cc \
  $(A)/compute/kernels/vector_selection.cc $(A)/compute/kernels/vector_sort.cc \
  $(A)/compute/kernels/vector_hash.cc \
  $(A)/compute/kernels/scalar_cast_boolean.cc $(A)/compute/kernels/scalar_cast_internal.cc \
  $(A)/compute/kernels/scalar_cast_numeric.cc

OBJS := $(patsubst $(REF)/%.cc,$(OBJ)/%.o,$(SRCS))

all: $(OUT)/libarrow_ref.so $(OUT)/libref_driver.so

$(OBJ)/%.o: $(REF)/%.cc
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) -c $< -o $@
	@objcopy --weaken $@

$(OUT)/libarrow_ref.so: $(OBJS)
	$(CXX) -shared -o $@ $(OBJS) -Wl,-z,lazy -Wl,--no-as-needed -lstdc++ -lm -lpthread -ldl

# The driver is OUR code (oracle/ref_driver.cc): a C-ABI over the reference's
# CallFunction / GroupBy with the same signatures as oracle/oracle.h.
$(OUT)/libref_driver.so: oracle/ref_driver.cc oracle/oracle.h $(OUT)/libarrow_ref.so
	$(CXX) $(CXXFLAGS) -Ioracle -shared -o $@ oracle/ref_driver.cc -L$(OUT) -larrow_ref \
	    -Wl,-z,lazy -Wl,--allow-shlib-undefined -Wl,-rpath,'$$ORIGIN'

.PHONY: all
