%% erlamsa_hip_batcher — the request side of SURVEY §8(f)-1: what erlamsa_esi:call_fuzzer/3 (erlamsa_esi.erl:86-95) and the
%% worker erlamsa_fsupervisor spawns per request (erlamsa_fsupervisor.erl:60-86) call instead of erlamsa_main:fuzzer/1.
%%
%% Every request process calls fuzz/3 and blocks; this server collects the requests that arrive within `hip_flush_us`
%% microseconds (default 200) or until `hip_batch` of them (default 4096) are waiting, runs them as ONE
%% erlamsa_hip:fuzz_calls/2 (= eh_fuzz_calls: every request keeps its own seed and is its own fuzzer/1 run with n = 1, so
%% the bytes are what erlamsa_app:fuzz(Bin, #{seed => Seed}) gives) and answers each caller.  A request that stopped at an
%% engine-only limit (NotRun) is run again on BEAM by its caller; a request the reference itself records nothing for
%% (empty result, dead worker) gets <<>> like erlamsa_main's FuzzingLoop gives it.
%%
%% Shipped as source like the NIF (this image has no OTP to compile it with).  The engine-side alternative without a
%% batcher process is erlamsa_hip:submit/3 + flush/1 + poll/2 (eh_submit / eh_flush / eh_poll).
-module(erlamsa_hip_batcher).
-behaviour(gen_server).

-export([start_link/1, fuzz/3]).
-export([init/1, handle_call/3, handle_cast/2, handle_info/2, terminate/2, code_change/3]).

-record(st, {dict, max, flush_ms, pending = [], n = 0, timer = undefined}).

%% Dict: the options map the service passes to erlamsa_main:fuzzer/1 (mutations, patterns, blockscale, ...), plus
%% hip_batch and hip_flush_us.
start_link(Dict) ->
    gen_server:start_link({local, ?MODULE}, ?MODULE, Dict, []).

%% -> {ok, binary()} | {rerun, Status}   (rerun: the engine could not finish this case; run erlamsa_main:fuzzer/1 for it)
fuzz(Bin, Seed, Timeout) when is_binary(Bin) ->
    gen_server:call(?MODULE, {fuzz, Bin, Seed}, Timeout).

init(Dict) ->
    FlushUs = maps:get(hip_flush_us, Dict, 200),
    {ok, #st{dict = Dict, max = maps:get(hip_batch, Dict, 4096), flush_ms = max(1, (FlushUs + 999) div 1000)}}.

handle_call({fuzz, Bin, Seed}, From, S = #st{pending = P, n = N, max = Max}) ->
    S1 = S#st{pending = [{From, Bin, Seed} | P], n = N + 1},
    case N + 1 >= Max of
        true -> {noreply, run(S1)};
        false -> {noreply, arm(S1)}
    end;
handle_call(_Other, _From, S) ->
    {reply, {error, badarg}, S}.

handle_cast(_Msg, S) ->
    {noreply, S}.

handle_info(flush, S) ->
    {noreply, run(S#st{timer = undefined})};
handle_info(_Other, S) ->
    {noreply, S}.

terminate(_Reason, #st{pending = P}) ->
    [gen_server:reply(From, {rerun, shutdown}) || {From, _, _} <- P],
    ok.

code_change(_Old, S, _Extra) ->
    {ok, S}.

%% the first request of a batch starts the clock (send_after has millisecond resolution: 200 us rounds up to 1 ms)
arm(S = #st{timer = undefined, flush_ms = Ms}) ->
    S#st{timer = erlang:send_after(Ms, self(), flush)};
arm(S) ->
    S.

run(S = #st{pending = []}) ->
    S;
run(S = #st{pending = P, dict = Dict, timer = T}) ->
    case T of
        undefined -> ok;
        _ ->
            %% the timer may have fired while a batch was running: its `flush` is in the mailbox already and would launch
            %% the NEXT batch early (and orphan that batch's own timer)
            erlang:cancel_timer(T, [{async, false}, {info, false}]),
            receive flush -> ok after 0 -> ok end
    end,
    Reqs = lists:reverse(P),                                   %% arrival order = case order of the batch
    Res = (catch erlamsa_hip:fuzz_calls([{B, Sd} || {_From, B, Sd} <- Reqs], Dict)),
    answer(Reqs, Res),
    S#st{pending = [], n = 0, timer = undefined}.

answer(Reqs, {ok, Outs, NotRun}) ->
    OutMap = maps:from_list(Outs),                             %% [{Index, Bin}]: recorded results (status 0, non-empty)
    NotMap = maps:from_list(NotRun),                           %% [{Index, Status}]: stopped at an engine-only limit
    lists:foldl(
        fun({From, _B, _Sd}, I) ->
            Reply = case maps:find(I, OutMap) of
                        {ok, Out} -> {ok, Out};
                        error ->
                            case maps:find(I, NotMap) of
                                {ok, St} -> {rerun, St};
                                error -> {ok, <<>>}            %% nothing recorded: record_result/2 drops <<>>
                            end
                    end,
            gen_server:reply(From, Reply),
            I + 1
        end, 1, Reqs),
    ok;
answer(Reqs, Error) ->                                         %% {error, _} from the NIF, or it is not loaded: everybody falls back
    [gen_server:reply(From, {rerun, Error}) || {From, _B, _Sd} <- Reqs],
    ok.
