// eh_otp_sort.h — OTP stdlib lists:sort/2 as the ENGINE restates it (host code; index cursors over arrays).
//
// erlamsa_utils:sort_by_priority/1 (reference src/erlamsa_utils.erl:113-117) hands lists:sort/2 a strict '>' as the
// ordering fun, so the order among equal priorities — which decides every weighted choice of mutator, pattern and
// generator — is whatever stdlib's merge sort does with a fun that is not a total "=<".  The engine's host set-up
// (eh_engine.hip) uses this header and nothing else does: oracle/otp_compat.h has a restatement of its own (cons lists, clause
// for clause from stdlib's lists.erl) and tests/pymodel.py a third one in Python; tests/test_pymodel.py
// (test_three_restatements_of_lists_sort_agree) diffs the three through eh_selftest_sort_by_priority.
#pragma once
#include <cstddef>
#include <functional>
#include <utility>
#include <vector>

namespace otp {

// ---------------------------------------------------------------------------
// lists:sort/2 — stdlib lists.erl (fsplit_* / fmergel / rfmergel family),
// restated 1:1 so that orderings with a non-total "=<" fun (erlamsa_utils.erl:115
// uses a strict '>') come out as on BEAM.
// ---------------------------------------------------------------------------
template <class T>
class ListsSort {
  using L = std::vector<T>;
  using F = std::function<bool(const T&, const T&)>;
  F fun;
  static L cons(const T& h, const L& t) {
    L r; r.reserve(t.size() + 1); r.push_back(h); r.insert(r.end(), t.begin(), t.end()); return r;
  }
  static L rev_onto(const L& a, const L& tail) {  // lists:reverse(A, Tail)
    L r(a.rbegin(), a.rend()); r.insert(r.end(), tail.begin(), tail.end()); return r;
  }
  // fmerge2_1(T1, H2, Fun, T2, M) etc. operate on index cursors to avoid copying.
  L fmerge2(const L& t1, const L& l2) {  // l2 = [H2|T2]; result is REVERSED (acc list M)
    L m; size_t i = 0, j = 0;
    // state 1: have H2=l2[j]; compare t1[i] with H2
    // fmerge2_1([H1|T1],H2,..): Fun(H1,H2) ? push H1 : (push H2 -> fmerge2_2)
    // fmerge2_2(H1,T1,Fun,[H2|T2],M): Fun(H1,H2) ? push H1 -> fmerge2_1 : push H2 -> fmerge2_2
    // Both states apply the same comparison rule; only the exhausted-list exits differ.
    while (true) {
      if (i == t1.size()) {  // fmerge2_1([], H2, _, T2, M) -> reverse(T2, [H2|M])
        L rest(l2.begin() + j, l2.end());
        // M currently reversed accumulation; result list = reverse(T2) ++ [H2|M]
        L out(rest.rbegin(), rest.rend());
        out.insert(out.end(), m.rbegin(), m.rend());
        return out;
      }
      if (j == l2.size()) {  // fmerge2_2(H1, T1, _, [], M) -> reverse(T1, [H1|M])
        L rest(t1.begin() + i, t1.end());
        L out(rest.rbegin(), rest.rend());
        out.insert(out.end(), m.rbegin(), m.rend());
        return out;
      }
      if (fun(t1[i], l2[j])) m.push_back(t1[i++]); else m.push_back(l2[j++]);
    }
  }
  L rfmerge2(const L& t1, const L& l2) {  // rfmerge2_1/2_2: Fun(H1,H2) ? push H2 : push H1
    L m; size_t i = 0, j = 0;
    while (true) {
      if (i == t1.size()) {
        L rest(l2.begin() + j, l2.end());
        L out(rest.rbegin(), rest.rend());
        out.insert(out.end(), m.rbegin(), m.rend());
        return out;
      }
      if (j == l2.size()) {
        L rest(t1.begin() + i, t1.end());
        L out(rest.rbegin(), rest.rend());
        out.insert(out.end(), m.rbegin(), m.rend());
        return out;
      }
      if (fun(t1[i], l2[j])) m.push_back(l2[j++]); else m.push_back(t1[i++]);
    }
  }
  // NOTE: m above is kept in push order; the Erlang M is the reverse of it, hence
  // out = reverse(rest) ++ reverse(m_pushorder)  ==  lists:reverse(Rest, [..|M]).
  L fmergel(std::vector<L> ls, std::vector<L> acc, bool asc);
  L rfmergel(std::vector<L> ls, std::vector<L> acc, bool asc);

 public:
  explicit ListsSort(F f) : fun(std::move(f)) {}
  L sort(const L& in);
};

template <class T>
typename ListsSort<T>::L ListsSort<T>::fmergel(std::vector<L> ls, std::vector<L> acc, bool asc) {
  // acc is a cons-list: new elements are pushed at the FRONT.
  while (true) {
    if (ls.size() >= 2) {
      if (asc) {  // fmergel([T1,[H2|T2]|L],Acc,Fun,asc)
        L merged = fmerge2(ls[0], ls[1]);
        acc.insert(acc.begin(), merged);
      } else {  // fmergel([[H2|T2],T1|L],Acc,Fun,desc)
        L merged = fmerge2(ls[1], ls[0]);
        acc.insert(acc.begin(), merged);
      }
      ls.erase(ls.begin(), ls.begin() + 2);
      continue;
    }
    if (ls.size() == 1) {
      if (acc.empty()) return ls[0];
      L r(ls[0].rbegin(), ls[0].rend());
      acc.insert(acc.begin(), r);
      return rfmergel(acc, {}, asc);
    }
    return rfmergel(acc, {}, asc);
  }
}

template <class T>
typename ListsSort<T>::L ListsSort<T>::rfmergel(std::vector<L> ls, std::vector<L> acc, bool asc) {
  while (true) {
    if (ls.size() >= 2) {
      if (asc) {  // rfmergel([[H2|T2],T1|L],Acc,Fun,asc)
        L merged = rfmerge2(ls[1], ls[0]);
        acc.insert(acc.begin(), merged);
      } else {  // rfmergel([T1,[H2|T2]|L],Acc,Fun,desc)
        L merged = rfmerge2(ls[0], ls[1]);
        acc.insert(acc.begin(), merged);
      }
      ls.erase(ls.begin(), ls.begin() + 2);
      continue;
    }
    if (ls.size() == 1) {
      L r(ls[0].rbegin(), ls[0].rend());
      acc.insert(acc.begin(), r);
      return fmergel(acc, {}, asc);
    }
    return fmergel(acc, {}, asc);
  }
}

template <class T>
typename ListsSort<T>::L ListsSort<T>::sort(const L& in) {
  if (in.size() < 2) return in;
  // Runs are cons-lists built by prepending; we keep them as vectors in list order.
  std::vector<L> rs;  // Rs (front = most recent)
  size_t pos = 2;
  T x = in[0], y = in[1];
  bool asc = fun(x, y);  // true -> fsplit_1, false -> fsplit_2
  // `ok(a,b)`: the comparison that continues the current run direction.
  auto step = [&](const T& a, const T& b) { return asc ? fun(a, b) : !fun(a, b); };
  L r;  // R as cons-list, front = most recently pushed
  bool have_s = false; T s{};
  while (true) {
    if (pos == in.size()) {
      // end of input
      L run; run.push_back(y); run.push_back(x); run.insert(run.end(), r.begin(), r.end());
      std::vector<L> all;
      if (have_s) all.push_back(L{s});
      all.push_back(run);
      all.insert(all.end(), rs.begin(), rs.end());
      return asc ? rfmergel(all, {}, true) : fmergel(all, {}, false);
    }
    const T z = in[pos++];
    if (step(y, z)) {            // Fun(Y,Z) continues the run: fsplit(Z, Y, [X|R])
      r.insert(r.begin(), x); x = y; y = z;
    } else if (step(x, z)) {     // fsplit(Y, Z, [X|R])
      r.insert(r.begin(), x); x = z;
    } else if (!have_s && r.empty()) {  // fsplit(Y, X, L, [Z], Rs)
      r.push_back(z);
    } else if (!have_s) {        // -> fsplit_x_1 with S = Z
      have_s = true; s = z;
    } else {                     // in fsplit_x_1: third comparison against S
      L run; run.push_back(y); run.push_back(x); run.insert(run.end(), r.begin(), r.end());
      rs.insert(rs.begin(), run);
      r.clear();
      if (step(s, z)) { y = z; x = s; } else { y = s; x = z; }
      have_s = false;
    }
  }
}


}  // namespace otp
