%% erlamsa_hip — drop-in batch path behind erlamsa_main:fuzzer/1 / erlamsa_app:fuzz/2.
%%
%% fuzz_batch/2 gives the same bytes as N worker iterations of erlamsa_main:fuzzer/1
%% (paths = [direct], workers = 1) where iteration I mutates the I-th binary.  Cases whose status
%% is not 0 (crashed = <<>> in the reference; overflow/unsupported/arena_full = engine limits) are
%% re-run on BEAM by the caller if it wants them.
%%
%% Also usable as an external module (`-e erlamsa_hip`, erlamsa_cmdparse.erl:456-470):
%% capabilities() -> {fuzzer, external}; fuzzer(Proto, Data, Opts) routes single packets of the
%% proxy through the GPU path (erlamsa_utils:make_fuzzer/1, erlamsa_utils.erl:221-226).
-module(erlamsa_hip).
-export([init/0, open/1, fuzz_batch/2, fuzz_batch_nif/5, capabilities/0, fuzzer/3]).
-on_load(init/0).

init() ->
    PrivDir = case code:priv_dir(erlamsa) of {error, _} -> "priv"; D -> D end,
    erlang:load_nif(filename:join(PrivDir, "erlamsa_hip_nif"), 0).

open(_Device) -> erlang:nif_error(nif_not_loaded).
fuzz_batch_nif(_Ctx, _Opts, _Seed, _FirstCase, _Bins) -> erlang:nif_error(nif_not_loaded).

%% Opts: the Dict of erlamsa_main:fuzzer/1 (seed, mutations, patterns, blockscale)
fuzz_batch(Bins, Dict) ->
    Ctx = case get(erlamsa_hip_ctx) of
              undefined -> {ok, C} = open(maps:get(device, Dict, 0)), put(erlamsa_hip_ctx, C), C;
              C -> C
          end,
    Seed = maps:get(seed, Dict, erlamsa_rnd:gen_urandom_seed()),
    Mutas = maps:get(mutations, Dict, erlamsa_mutations:default([])),
    Pats = maps:get(patterns, Dict, erlamsa_patterns:default()),
    {SsrfHost, SsrfPort} = erlamsa_mutations:get_ssrf_ep(),
    Opts = #{mutations => actions(Mutas), patterns => actions(Pats),
             blockscale => float(maps:get(blockscale, Dict, 1.0)),
             ssrf_host => SsrfHost, ssrf_port => SsrfPort},
    case fuzz_batch_nif(Ctx, Opts, Seed, maps:get(first_case, Dict, 1), Bins) of
        {ok, Res} -> [Bin || {0, Bin} <- Res, Bin =/= <<>>];   %% record_result/2 drops <<>>
        {error, Why} -> {error, Why}                            %% caller falls back to erlamsa_main:fuzzer/1
    end.

actions(L) -> string:join([atom_to_list(N) ++ "=" ++ integer_to_list(P) || {N, P} <- L], ",").

capabilities() -> {fuzzer, external}.
fuzzer(_Proto, Data, Opts) ->
    case fuzz_batch([Data], Opts) of [Out] -> {ok, Out}; _ -> {ok, Data} end.
