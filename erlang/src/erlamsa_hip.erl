%% erlamsa_hip — drop-in batch path behind erlamsa_main:fuzzer/1 / erlamsa_app:fuzz/2.
%%
%% fuzz_batch/2 gives the bytes N worker iterations of erlamsa_main:fuzzer/1 (paths = [direct], workers = 1) give,
%% iteration I mutating the I-th binary; fuzz_calls/2 gives what N separate erlamsa_app:fuzz(Bin, #{seed => S}) calls
%% give.  Both return
%%     {ok, Outs :: [{Index, binary()}], NotRun :: [{Index, Status}]}
%% Outs    the cases the reference would have recorded (status 0 and a non-empty result: erlamsa_main record_result/2
%%         drops <<>>, which is also what a crashed worker — status 1 — yields), in case order, with their positions;
%% NotRun  the cases that stopped at an ENGINE limit the reference does not have (2 overflow of big_case_bytes,
%%         3 unsupported, 4 output arena full, 5 work budget).  The reference would have produced output for them: the
%%         caller re-runs them on BEAM (or with larger limits) — nothing is dropped silently.
%%
%% Also usable as an external module (`-e erlamsa_hip`, erlamsa_cmdparse.erl:456-470): capabilities() ->
%% {fuzzer, external}; fuzzer(Proto, Data, Opts) routes single packets of the proxy through the GPU path
%% (erlamsa_utils:make_fuzzer/1, erlamsa_utils.erl:221-226) and falls back to the unmodified packet.
-module(erlamsa_hip).
-export([init/0, open/1, fuzz_batch/2, fuzz_calls/2, fuzz_batch_nif/5, fuzz_calls_nif/4, capabilities/0, fuzzer/3]).
-export([submit/3, flush/1, poll/2, submit_nif/4, flush_nif/1, poll_nif/2]).
-export([write_files/3, write_files_nif/3]).
-export([meta_terms/1, meta_terms/3, meta_print/2, meta_nif/2, meta_atoms_nif/0]).
-export([fuzz_batch_multi/2, open_all/0, device_count/0, load_corpus_nif/2, fuzz_range_nif/6, comm_init_local_nif/1, broadcast_local_nif/2,
         comm_unique_id_nif/0, comm_init_nif/4, corpus_broadcast_nif/3, case_range/3, join_ranks/3]).
-on_load(init/0).

init() ->
    PrivDir = case code:priv_dir(erlamsa) of {error, _} -> "priv"; D -> D end,
    erlang:load_nif(filename:join(PrivDir, "erlamsa_hip_nif"), 0).

open(_Device) -> erlang:nif_error(nif_not_loaded).
fuzz_batch_nif(_Ctx, _Opts, _Seed, _FirstCase, _Bins) -> erlang:nif_error(nif_not_loaded).
fuzz_calls_nif(_Ctx, _Opts, _Seeds, _Bins) -> erlang:nif_error(nif_not_loaded).
submit_nif(_Ctx, _Opts, _Seed, _Bin) -> erlang:nif_error(nif_not_loaded).
flush_nif(_Ctx) -> erlang:nif_error(nif_not_loaded).
poll_nif(_Ctx, _Ticket) -> erlang:nif_error(nif_not_loaded).
write_files_nif(_Ctx, _Template, _FirstN) -> erlang:nif_error(nif_not_loaded).
meta_nif(_Ctx, _I) -> erlang:nif_error(nif_not_loaded).
meta_atoms_nif() -> erlang:nif_error(nif_not_loaded).
device_count() -> erlang:nif_error(nif_not_loaded).
load_corpus_nif(_Ctx, _Bins) -> erlang:nif_error(nif_not_loaded).
fuzz_range_nif(_Ctx, _Opts, _Seed, _FirstCase, _CorpusFirst, _N) -> erlang:nif_error(nif_not_loaded).
comm_init_local_nif(_Ctxs) -> erlang:nif_error(nif_not_loaded).
broadcast_local_nif(_Ctxs, _Root) -> erlang:nif_error(nif_not_loaded).
comm_unique_id_nif() -> erlang:nif_error(nif_not_loaded).
comm_init_nif(_Ctx, _Id, _Rank, _N) -> erlang:nif_error(nif_not_loaded).
corpus_broadcast_nif(_Ctx, _Root, _BinsOrNone) -> erlang:nif_error(nif_not_loaded).

%% Dict: the options map of erlamsa_main:fuzzer/1 (seed, mutations, patterns, generators, blockscale) plus first_case / device
fuzz_batch(Bins, Dict) ->
    case host_only(Dict) of
        [] ->
            Seed = maps:get(seed, Dict, erlamsa_rnd:gen_urandom_seed()),
            First = maps:get(first_case, Dict, 1),
            split(fuzz_batch_nif(ctx(Dict), opts(Dict), Seed, First, Bins), First, maps:get(skip, Dict, 0));
        Keys -> {error, {unsupported, Keys}}                      %% caller falls back to erlamsa_main:fuzzer/1
    end.

%% Keys of the Dict that erlamsa_main:fuzzer/1 honours and a batch on the GPU cannot: custom mutators of an external module appended
%% to the table (external_mutations: erlamsa_main.erl:128, erlamsa_mutations.erl:1332, Erlang funs of fun/2 type) and the
%% per-block post-processor (external_post: erlamsa_main.erl:159, applied to every block as erlamsa_out:blocks_port/5 writes it -
%% the engine hands over the case's bytes as ONE binary, the block boundaries are gone).  A run that carries one of them is
%% refused, like sequence_muta, instead of running WITHOUT what the user asked for.
host_only(Dict) ->
    [K || K <- [external_mutations, external_post], maps:get(K, Dict, nil) =/= nil].

%% ---- the meta trace (-M, erlamsa_main.erl:58-70) ---------------------------------------------------------------------
%% With #{meta => true} in Dict the engine keeps every case's Meta list (EH_FLAG_META_TRACE).  meta_nif(Ctx, I) hands over case I's
%% event bytes (0-based index into the last batch), meta_terms/1 turns them into the reference's OWN terms in the order its meta
%% logger prints them - [{pattern,once_dec},{byte_drop,-1},{used,bd}] ... - and meta_terms/3 adds what the host side of fuzzer/1
%% conses around Pat(Ll, Muta, Meta): {nth, I}, the generator's entry, the output's, {written, N}.  meta_print/2 is the logger:
%% every element with ~p on a line of its own, exactly the text erlamsa -M writes.
meta_terms(Bytes) ->
    %% the engine keeps 32 KiB per case and marks a cut with a last byte 16#FF - wherever the cut fell, also in the middle of an event
    {Body, Cut} = case Bytes of
                      <<B:32767/binary, 16#FF>> -> {B, true};
                      _ -> {Bytes, false}
                  end,
    Terms = meta_decode(Body, list_to_tuple(meta_atoms()), []),
    case Cut andalso (Terms =:= [] orelse lists:last(Terms) =/= {meta, truncated}) of
        true -> Terms ++ [{meta, truncated}];
        false -> Terms
    end.
meta_terms(Bytes, {Nth, GenMeta, OutMeta}, Written) ->
    %% [{nth, I}, GenMeta] ++ the output's entry (erlamsa_main.erl:185-186) in front, {written, N} (:195) behind; GenMeta may be a
    %% list ([{generator, file}, {source, path}], erlamsa_gen.erl:115): lists:flatten keeps its order, the final reverse turns it round
    [{nth, Nth}] ++ lists:reverse(lists:flatten([GenMeta])) ++ [OutMeta] ++ meta_terms(Bytes) ++ [{written, Written}].
meta_print(Verb, Terms) -> lists:foreach(fun(X) -> Verb(io_lib:format("~p~n", [X])) end, Terms).

meta_atoms() ->
    case persistent_term:get(erlamsa_hip_meta_atoms, undefined) of
        undefined -> A = meta_atoms_nif(), persistent_term:put(erlamsa_hip_meta_atoms, A), A;
        A -> A
    end.
%% events: include/erlamsa_hip.h eh_result_meta (csrc/eh_common.h TraceKind)
meta_decode(Bin, At, Acc) ->
    %% an event the 32 KiB limit cut in two (operands missing, a varint without its last byte, an atom id beyond the table) ends
    %% the list as {meta, truncated} instead of raising
    try meta_event(Bin, At, Acc) catch error:_ -> lists:reverse([{meta, truncated} | Acc]) end.
meta_event(<<>>, _At, Acc) -> lists:reverse(Acc);
meta_event(<<16#FF>>, _At, Acc) -> lists:reverse([{meta, truncated} | Acc]);
meta_event(<<1, A, B, R/binary>>, At, Acc) -> meta_decode(R, At, [{element(A + 1, At), element(B + 1, At)} | Acc]);
meta_event(<<2, A, R0/binary>>, At, Acc) -> {Z, R} = varint(R0), meta_decode(R, At, [{element(A + 1, At), (Z bsr 1) bxor -(Z band 1)} | Acc]);
meta_event(<<3, S8, Big, R0/binary>>, At, Acc) ->
    {Len, R1} = varint(R0), {A, R2} = varint(R1), {B, R} = varint(R2),
    Endian = case Big of 1 -> big; 0 -> little end,
    meta_decode(R, At, [{sizer, {ok, S8 * 8, Endian, Len, A, B}} | Acc]);
meta_event(<<4, Crc, R0/binary>>, At, Acc) ->
    {PLen, R1} = varint(R0), {BLen, R} = varint(R1),
    E = case Crc of 1 -> {crc32, 32, PLen, BLen}; 0 -> {xor8, 8, PLen, BLen} end,
    meta_decode(R, At, [{csum, E} | Acc]);
meta_event(<<5, R0/binary>>, At, Acc) -> {N, R} = varint(R0), meta_decode(R, At, [{skipped, N * 8 / 8} | Acc]);   %% Len/8, Len in bits (erlamsa_patterns.erl:152-154)
meta_event(<<6, R0/binary>>, At, Acc) -> {N, R1} = varint(R0), <<Name:N/binary, R/binary>> = R1, meta_decode(R, At, [{archiver, binary_to_list(Name)} | Acc]);
meta_event(_Cut, _At, Acc) -> lists:reverse([{meta, truncated} | Acc]).     %% the engine keeps 32 KiB per case: an event cut by that limit
varint(B) -> varint(B, 0, 0).
varint(<<1:1, V:7, R/binary>>, S, Acc) -> varint(R, S + 7, Acc bor (V bsl S));
varint(<<0:1, V:7, R/binary>>, S, Acc) -> {Acc bor (V bsl S), R}.

%% ---- several GPUs -------------------------------------------------------------------------------------------------------
%% One BEAM node, every GPU of the machine: the arena is uploaded to the first device, RCCL-broadcast to the others from inside
%% the library (no HIP binding needed here), and every device runs its contiguous range of the case numbers - the shape of
%% erlamsa_main:get_threading_mode/3 (erlamsa_main.erl:95-108), except that results do NOT depend on the number of devices
%% (case I is parent draws s0+3(I-1)+1.., whoever runs it).  Same return value as fuzz_batch/2.
fuzz_batch_multi(Bins, Dict) ->
    case host_only(Dict) of
        [] ->
            %% load, broadcast and the per-device ranges are ONE critical section of the node: the contexts are shared by every
            %% process (persistent_term), and a second caller's corpus must not replace this one's between its load and its ranges
            global:trans({erlamsa_hip_multi_run, self()}, fun() -> fuzz_batch_multi_locked(Bins, Dict) end, [node()]);
        Keys -> {error, {unsupported, Keys}}
    end.
fuzz_batch_multi_locked(Bins, Dict) ->
    Ctxs = multi_ctxs(),
    W = length(Ctxs), N = length(Bins),
    Seed = maps:get(seed, Dict, erlamsa_rnd:gen_urandom_seed()),
    First = maps:get(first_case, Dict, 1),
    Opts = opts(Dict),
    ok = load_corpus_nif(hd(Ctxs), Bins),
    ok = case W of 1 -> ok; _ -> broadcast_local_nif(Ctxs, 0) end,
    Parent = self(),
    Pids = [spawn_link(fun() -> {A, Cnt} = case_range(N, R, W),
                                Parent ! {self(), fuzz_range_nif(C, Opts, Seed, First + A, A, Cnt)} end)
            || {R, C} <- lists:zip(lists:seq(0, W - 1), Ctxs)],
    Parts = [receive {P, Res} -> Res end || P <- Pids],
    case [E || {error, _} = E <- Parts] of
        [E | _] -> E;                                             %% caller falls back to erlamsa_main:fuzzer/1
        [] -> split({ok, lists:append([L || {ok, L} <- Parts])}, First, maps:get(skip, Dict, 0))
    end.

%% contiguous split of cases 0..N-1 over W ranks, the first N rem W ranks take one more (erlamsa_amd/shard.py case_range)
case_range(N, Rank, W) ->
    Base = N div W, Rem = N rem W,
    {Rank * Base + min(Rank, Rem), Base + case Rank < Rem of true -> 1; false -> 0 end}.

open_all() -> [begin {ok, C} = open(D), C end || D <- lists:seq(0, device_count() - 1)].
multi_ctxs() ->
    case persistent_term:get(erlamsa_hip_multi, undefined) of
        undefined ->
            global:trans({erlamsa_hip_multi, self()},
                         fun() ->
                             case persistent_term:get(erlamsa_hip_multi, undefined) of
                                 undefined ->
                                     Cs = open_all(),
                                     ok = case Cs of [_] -> ok; _ -> comm_init_local_nif(Cs) end,
                                     persistent_term:put(erlamsa_hip_multi, Cs), Cs;
                                 Cs -> Cs
                             end
                         end, [node()]);
        Cs -> Cs
    end.

%% One BEAM node PER GPU (a cluster of nodes on one machine, or `--workers` across nodes): rank 0 makes the unique id, Erlang
%% distribution carries its 128 bytes, every node joins, the root's Bins reach all of them over xGMI.
%%   Nodes :: [node()] in rank order, this node among them.  -> {ok, Ctx, Rank}
%% The id travels by request and reply: every other rank asks rank 0 (again every 200 ms until it answers - rank 0 may not have
%% registered yet, and a message to a name nobody holds is dropped), rank 0 answers the N - 1 requests.  Nobody enters the
%% collective ncclCommInitRank before all ranks hold the id; a rank that does not hear from the others within
%% join_timeout ms (default 60 000) returns {error, timeout} instead of blocking a dirty scheduler for good.
join_ranks(Nodes, Bins, Dict) ->
    Rank = length(lists:takewhile(fun(Nd) -> Nd =/= node() end, Nodes)),
    Timeout = maps:get(join_timeout, Dict, 60000),
    case join_id(Rank, Nodes, Timeout) of
        {ok, Id} ->
            C = ctx(Dict),
            ok = comm_init_nif(C, Id, Rank, length(Nodes)),
            ok = corpus_broadcast_nif(C, 0, case Rank of 0 -> Bins; _ -> none end),
            {ok, C, Rank};
        Error -> Error
    end.

join_id(0, Nodes, Timeout) ->
    {ok, Id} = comm_unique_id_nif(),
    try register(erlamsa_hip_uid_root, self()) of
        true ->
            Res = serve_id(Id, length(Nodes) - 1, erlang:monotonic_time(millisecond) + Timeout),
            unregister(erlamsa_hip_uid_root),
            Res
    catch error:badarg -> {error, join_in_progress}               %% a second join on this node while the first one runs
    end;
join_id(_Rank, Nodes, Timeout) -> ask_id(hd(Nodes), erlang:monotonic_time(millisecond) + Timeout).

serve_id(Id, 0, _Deadline) -> {ok, Id};
serve_id(Id, Left, Deadline) ->
    receive {uid_req, From} -> From ! {uid, Id}, serve_id(Id, Left - 1, Deadline)
    after max(0, Deadline - erlang:monotonic_time(millisecond)) -> {error, timeout}
    end.
ask_id(Root, Deadline) ->
    {erlamsa_hip_uid_root, Root} ! {uid_req, self()},
    receive {uid, Id} -> {ok, Id}
    after 200 ->
        case erlang:monotonic_time(millisecond) >= Deadline of
            true -> {error, timeout};
            false -> ask_id(Root, Deadline)                           %% (a second answer to a repeated request is the same id: harmless)
        end
    end.

%% Calls :: [{Bin, Seed}] — one erlamsa_app:fuzz(Bin, #{seed => Seed}) each
fuzz_calls(Calls, Dict) ->
    case host_only(Dict) of
        [] ->
            {Bins, Seeds} = lists:unzip(Calls),
            split(fuzz_calls_nif(ctx(Dict), opts(Dict), Seeds, Bins), 1, 0);   %% (every call is case 1 of its own run: skip => N >= 1 would drop them all, as the reference does)
        Keys -> {error, {unsupported, Keys}}
    end.

%% Request coalescing for erlamsa_fsupervisor-style services: every request process submits and then polls; a timer
%% process calls flush/1 every ~200 us (a full batch launches itself).  poll/2 -> {ok, Status, Bin} | again.
%% `-o "out-%n.bin"` for a batch: the files of the cases fuzz_batch/2 has just produced on this node's context, named like
%% erlamsa_out:file_writer/1 names them (erlamsa_out.erl:103-123) -> {ok, Files, Bytes, NotWritten}.
write_files(Template, FirstN, Dict) -> write_files_nif(ctx(Dict), Template, FirstN).

%% The coalescer has a context of its own: the engine refuses batches, corpora and configurations on a context with
%% requests pending ({error, wrong_call_order}), because they would overwrite what those requests run with.
submit(Bin, Seed, Dict) ->
    case host_only(Dict) of
        [] -> submit_nif(co_ctx(Dict), opts(Dict), Seed, Bin);
        Keys -> {error, {unsupported, Keys}}
    end.
flush(Dict) -> flush_nif(co_ctx(Dict)).
poll(Ticket, Dict) -> poll_nif(co_ctx(Dict), Ticket).

co_ctx(#{hip_co_ctx := C}) -> C;
co_ctx(Dict) -> shared_ctx(erlamsa_hip_co_ctx, Dict).

%% One GPU context per node, shared by all processes (the NIF serialises batches on it; coalescing needs the
%% requests of different processes in the same context).  #{hip_ctx => C} overrides it.
ctx(#{hip_ctx := C}) -> C;
ctx(Dict) -> shared_ctx(erlamsa_hip_ctx, Dict).

%% one context per key and node: two first callers must not open one each (check-then-put is not atomic), so the opening
%% is serialised through a global lock; later callers only read the persistent term
shared_ctx(Key, Dict) ->
    case persistent_term:get(Key, undefined) of
        undefined ->
            global:trans({Key, self()},
                         fun() ->
                             case persistent_term:get(Key, undefined) of
                                 undefined -> {ok, C} = open(maps:get(device, Dict, 0)), persistent_term:put(Key, C), C;
                                 C -> C
                             end
                         end, [node()]);
        C -> C
    end.

opts(Dict) ->
    Mutas = maps:get(mutations, Dict, erlamsa_mutations:default([])),
    Pats = maps:get(patterns, Dict, erlamsa_patterns:default()),
    {SsrfHost, SsrfPort} = erlamsa_mutations:get_ssrf_ep(),
    Base = #{mutations => actions(Mutas), patterns => actions(Pats),
             blockscale => float(maps:get(blockscale, Dict, 1.0)),
             ssrf_host => SsrfHost, ssrf_port => SsrfPort},
    %% generators => [{Name, Pri}] as erlamsa_main:fuzzer/1 takes them (erlamsa_gen:default/0); `file` and `jump` stream the
    %% Bins of the batch as their Paths on the device (erlamsa_gen.erl:106-150), stdin / genfuz stay on BEAM
    Gens = case maps:find(generators, Dict) of
               {ok, G} -> #{generators => actions([{N, P} || {N, P} <- G, lists:member(N, [direct, random, file, jump])])};
               error -> #{}
           end,
    %% sequence_muta (--consequtive-mutators, erlamsa_main.erl:223-235) goes along as it is: the engine refuses it
    %% ({error, "sequence_muta ..."}) and the caller's fall-back to erlamsa_main:fuzzer/1 takes the run
    maps:merge(maps:merge(Base, Gens), maps:with([max_case_bytes, big_case_bytes, max_case_work, sequence_muta, meta], Dict)).

%% First: the case number of the first result; Skip: the Dict's skip => N (erlamsa_main.erl:161,191-196): cases numbered <= N are
%% processed - their draws happen, the cases behind them are what they would be - but written to the `skip` port, which keeps
%% nothing (erlamsa_out.erl:677): their data is <<>> and record_result/2 drops it
split({error, Why}, _First, _Skip) -> {error, Why};              %% caller falls back to erlamsa_main:fuzzer/1
split({ok, Res}, First, Skip) ->
    Indexed = lists:zip(lists:seq(1, length(Res)), Res),
    Outs = [{I, Bin} || {I, {0, Bin}} <- Indexed, Bin =/= <<>>, First + I - 1 > Skip],  %% record_result/2 drops <<>>
    NotRun = [{I, St} || {I, {St, _}} <- Indexed, St >= 2, First + I - 1 > Skip],
    {ok, Outs, NotRun}.

actions(L) -> string:join([atom_to_list(N) ++ "=" ++ integer_to_list(P) || {N, P} <- L], ",").

capabilities() -> {fuzzer, external}.
fuzzer(_Proto, Data, Opts) ->
    case fuzz_batch([Data], Opts) of {ok, [{1, Out}], []} -> {ok, Out}; _ -> {ok, Data} end.
