"""erlamsa_amd — MI355X-native batch mutation engine behind erlamsa's API.

Host-side mirror (Python) of the reference's entry points for the one
accelerated path (batch fuzz-case generation):

    erlamsa_main:fuzzer(Dict)          -> fuzzer(opts)         (reference src/erlamsa_main.erl:124-247)
    erlamsa_app:fuzz(Bin[, Opts])      -> fuzz(data, opts)     (reference src/erlamsa_app.erl:255-263)
    batched form used by the NIF shim  -> fuzz_batch(inputs, opts)

All three call the C ABI of liberlamsa_hip.so (include/erlamsa_hip.h); there is
no CPU fallback.
"""
from .engine import (CASE_CRASHED, CASE_OK, CASE_OVERFLOW, CASE_UNSUPPORTED, Engine, EngineError, gpu_mutators,
                     gpu_patterns, load_library, mutator_table, pattern_table)
from .api import EngineLimit, actions_to_string, default_mutations, default_patterns, fuzz, fuzz_batch, fuzzer, pack_corpus

__all__ = ["Engine", "EngineError", "EngineLimit", "fuzzer", "fuzz", "fuzz_batch", "pack_corpus", "default_mutations",
           "default_patterns", "actions_to_string", "mutator_table", "pattern_table", "gpu_mutators", "gpu_patterns",
           "load_library", "CASE_OK", "CASE_CRASHED", "CASE_OVERFLOW", "CASE_UNSUPPORTED"]
