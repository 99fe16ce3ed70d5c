#!/usr/bin/env escript
%%! -noshell
%% capture.escript — run the REAL erlamsa over the golden inputs and print what it produces.
%%
%% This repository's image has no Erlang/OTP, so tests/golden/vectors.json is made by the CPU oracle (a C++ restatement)
%% and oracle-vs-BEAM parity is unpinned.  Anyone with OTP and a built erlamsa turns it into "pinned" with:
%%
%%     python tests/golden/make_golden.py                  # writes vectors.json and vectors.eterm (the same inputs)
%%     escript tests/golden/capture.escript /path/to/erlamsa/ebin tests/golden/vectors.eterm > tests/golden/beam_capture.txt
%%     python tests/golden/apply_beam_capture.py tests/golden/beam_capture.txt
%%     python -m pytest tests/test_golden.py               # the oracle (and, with -m gpu, the engine) against BEAM output
%%
%% Case I of a vector (1-based, first_case + I - 1 in the run's numbering) mutates inputs[I].  erlamsa_main:fuzzer/1 takes
%% one input per run, so the case is run as iteration N = that number with skip = N - 1: the skipped iterations still draw
%% their ThreadSeed from the parent stream (erlamsa_main.erl:179 precedes the skip test at :224), so iteration N gets
%% exactly the seed it has in a batch.  maxrunningtime is raised so that no case is cut by the 30 ms default watchdog of
%% the `return` output (erlamsa_out.erl:583-586): the captured bytes must not depend on this machine's speed.
main([Ebin, File]) ->
    true = code:add_patha(Ebin),
    {ok, [Vectors]} = file:consult(File),
    io:format("otp ~s~n", [erlang:system_info(otp_release)]),
    lists:foreach(fun run_vector/1, Vectors);
main(_) ->
    io:format(standard_error, "usage: capture.escript EBIN_DIR vectors.eterm > beam_capture.txt~n", []),
    halt(2).

run_vector({Name, Seed, First, Muts, Pats, Inputs}) ->
    B0 = #{paths => [direct], output => return, seed => Seed, maxrunningtime => 3600000},
    B1 = case Muts of default -> B0; _ -> B0#{mutations => actions(Muts, erlamsa_mutations:default([]))} end,
    B2 = case Pats of default -> B1; _ -> B1#{patterns => actions(Pats, erlamsa_patterns:default())} end,
    lists:foldl(
        fun(Hex, I) ->
            N = First + I,
            Res = (catch erlamsa_main:fuzzer(B2#{input => unhex(Hex), n => N, skip => N - 1})),
            %% record_result/2 drops <<>>, and a dead worker also yields <<>>: both print as an empty result
            Out = case Res of [B] when is_binary(B) -> B; _ -> <<>> end,
            io:format("~s ~p ~s~n", [Name, I, hex(Out)]),
            I + 1
        end, 0, Inputs).

%% "bd,bf=4,num" -> [{bd, 1}, {bf, 4}, {num, 3}]: a name without "=" takes its default priority
actions(Str, Defaults) ->
    [begin
         case string:tokens(Tok, "=") of
             [Name] -> A = list_to_atom(Name), {A, proplists:get_value(A, Defaults)};
             [Name, Pri] -> {list_to_atom(Name), list_to_integer(Pri)}
         end
     end || Tok <- string:tokens(Str, ",")].

unhex(L) -> unhex(L, <<>>).
unhex([], Acc) -> Acc;
unhex([A, B | T], Acc) -> unhex(T, <<Acc/binary, (list_to_integer([A, B], 16))>>).

hex(Bin) -> lists:flatten([io_lib:format("~2.16.0b", [X]) || <<X>> <= Bin]).
