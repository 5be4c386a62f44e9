#include <ctime>
// util, logsumexp, fastseq, alignpath and model parts of the host mirror (see hx_host.h).
#include "hx_host.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <iomanip>
#include <sstream>

namespace historian {

thread_local FillTiming fillTiming;
double wallSeconds() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}


// ---- diagnostics (behaviour of reference src/util.cpp:26-54: Warn continues, Abort terminates, Fail exits 1) -------
namespace {
enum Severity { kWarn, kAbort, kFail };
void report(Severity sev, const char* format, va_list args) {
  static const char* const label[] = {"Warning: ", "Abort: ", ""};
  std::string line(label[sev]);
  char text[2048];
  vsnprintf(text, sizeof(text), format, args);
  line += text;
  line += '\n';
  fputs(line.c_str(), stderr);       // one write per message: messages of concurrent device threads do not interleave
}
}  // namespace

void Warn(const char* warning, ...) {
  va_list args;
  va_start(args, warning);
  report(kWarn, warning, args);
  va_end(args);
}

void Abort(const char* error, ...) {
  va_list args;
  va_start(args, error);
  report(kAbort, error, args);
  va_end(args);
  std::terminate();   // (the reference reaches std::terminate through a `throw;` without an active exception)
}

void Fail(const char* error, ...) {
  va_list args;
  va_start(args, error);
  report(kFail, error, args);
  va_end(args);
  exit(EXIT_FAILURE);
}

// ---- log-sum-exp table (semantics of reference src/logsumexp.cpp:6-53) ------------------------------------------
// One entry more than the reference allocates: its interpolation reads lookup[n + 1] for n up to ENTRIES - 1.
LogSumExpLookupTable logSumExpLookupTable;

LogSumExpLookupTable::LogSumExpLookupTable() : lookup(new double[LOG_SUM_EXP_LOOKUP_ENTRIES + 1]) {
  double* entry = lookup;
  for (int bin = 0; bin <= LOG_SUM_EXP_LOOKUP_ENTRIES; ++bin) *entry++ = log_sum_exp_unary_slow(bin * LOG_SUM_EXP_LOOKUP_PRECISION);
}
LogSumExpLookupTable::~LogSumExpLookupTable() { delete[] lookup; }

// log(1 + e^-x) by libm, the value the table samples
double log_sum_exp_unary_slow(double x) { return log(1. + exp(-x)); }

double log_sum_exp_slow(double a, double b) {
  const double hi = a < b ? b : a, lo = a < b ? a : b;
  return std::isinf(lo) && lo < 0 ? hi : hi + log_sum_exp_unary_slow(hi - lo);
}
double log_sum_exp_slow(double a, double b, double c) { return log_sum_exp_slow(log_sum_exp_slow(a, b), c); }
double log_sum_exp_slow(double a, double b, double c, double d) { return log_sum_exp_slow(log_sum_exp_slow(a, b, c), d); }
void log_accum_exp_slow(double& a, double b) { a = log_sum_exp_slow(a, b); }

vguard<LogProb> log_vector(const vguard<double>& v) {
  vguard<LogProb> out;
  out.reserve(v.size());
  for (double p : v) out.push_back(log(p));
  return out;
}

// ---- sequences ------------------------------------------------------------------------------------------------
// index of a character in the alphabet, trying the other letter case second (reference src/fastseq.cpp:10-16)
UnvalidatedAlphTok tokenize(char c, const string& alphabet) {
  size_t at = alphabet.find(c);
  if (at == string::npos) at = alphabet.find((char)(isupper((unsigned char)c) ? tolower((unsigned char)c) : toupper((unsigned char)c)));
  return at == string::npos ? InvalidAlphabetToken : (UnvalidatedAlphTok)at;
}

vguard<FastSeq> readFastSeqs(const char* filename) {
  std::ifstream in(filename);
  Require(in.good(), "Couldn't open %s", filename);
  vguard<FastSeq> seqs;
  string line;
  while (std::getline(in, line)) {
    while (!line.empty() && (line.back() == '\r' || line.back() == '\n' || line.back() == ' ')) line.pop_back();
    if (line.empty()) continue;
    if (line[0] == '>') {
      FastSeq fs;
      const size_t sp = line.find_first_of(" \t");
      fs.name = line.substr(1, sp == string::npos ? string::npos : sp - 1);
      if (sp != string::npos) fs.comment = line.substr(sp + 1);
      seqs.push_back(fs);
    } else if (!seqs.empty()) {
      for (char c : line)
        if (!isspace((unsigned char)c)) seqs.back().seq.push_back(c);
    }
  }
  return seqs;
}

// ---- alignment paths (semantics of reference src/alignpath.cpp:6-81, 282-318) -------------------------------------
// An AlignPath is a set of rows of equal length; every operation below first asks for that width.
const char Alignment::gapChar = '-';
const char Alignment::wildcardChar = '*';

AlignColIndex alignPathColumns(const AlignPath& a) {
  if (a.empty()) return 0;
  const auto& ref = *a.begin();
  for (const auto& row : a)
    Assert(row.second.size() == ref.second.size(), "Alignment path is not flush: row %u has %u columns, but row %u has %u columns",
           (unsigned)ref.first, (unsigned)ref.second.size(), (unsigned)row.first, (unsigned)row.second.size());
  return ref.second.size();
}

SeqIdx alignPathResiduesInRow(const AlignRowPath& r) { return (SeqIdx)std::count(r.begin(), r.end(), true); }

// rows of both; a row present in both keeps the first argument's version
AlignPath alignPathUnion(const AlignPath& a1, const AlignPath& a2) {
  AlignPath merged(a2);
  for (const auto& row : a1) merged[row.first] = row.second;
  return merged;
}

// columns of a1 followed by the columns of a2; a row missing on one side is gaps there
AlignPath alignPathConcat(const AlignPath& a1, const AlignPath& a2) {
  const AlignColIndex w1 = alignPathColumns(a1), w2 = alignPathColumns(a2);
  AlignPath joined;
  for (const auto& row : a1) {
    AlignRowPath& out = joined[row.first];
    out = row.second;
    out.resize(w1 + w2, false);
  }
  for (const auto& row : a2) {
    AlignRowPath& out = joined[row.first];
    out.resize(w1, false);                   // (a new row: all gaps so far; an existing one: cut back to a1's part)
    out.insert(out.end(), row.second.begin(), row.second.end());
  }
  return joined;
}

AlignPath alignPathConcat(const AlignPath& a1, const AlignPath& a2, const AlignPath& a3) {
  return alignPathConcat(alignPathConcat(a1, a2), a3);
}

void ensureAlignPathHasRow(AlignPath& a, AlignRowIndex r) {
  const AlignColIndex width = alignPathColumns(a);
  a.emplace(r, AlignRowPath(width, false));  // (no effect when the row exists)
}

string alignPathString(const AlignPath& a) {
  std::ostringstream text;
  for (const auto& row : a) {
    text << std::setw(4) << row.first << ' ';
    for (bool residue : row.second) text << (residue ? Alignment::wildcardChar : Alignment::gapChar);
    text << std::endl;
  }
  return text.str();
}

// The band of a pair DP (reference src/alignpath.cpp:282-310): per guide column the number of columns so far in which
// both rows have a residue, and per sequence position of either row the column it sits in (position 0 = before the
// first residue = column 0).
GuideAlignmentEnvelope::GuideAlignmentEnvelope(const AlignPath& guide, AlignRowIndex row1, AlignRowIndex row2, int maxDistance)
    : row1(row1), row2(row2), maxDistance(maxDistance) {
  const auto r1 = guide.find(row1), r2 = guide.find(row2);
  Assert(r1 != guide.end(), "Guide alignment is missing row #%u", (unsigned)row1);
  Assert(r2 != guide.end(), "Guide alignment is missing row #%u", (unsigned)row2);
  const AlignColIndex width = alignPathColumns(guide);
  const AlignRowPath& in1 = r1->second;
  const AlignRowPath& in2 = r2->second;
  cumulativeMatches.assign(width + 1, 0);
  row1PosToCol.assign(1, 0);
  row2PosToCol.assign(1, 0);
  for (AlignColIndex col = 1; col <= width; ++col) {
    const bool has1 = in1[col - 1], has2 = in2[col - 1];
    if (has1) row1PosToCol.push_back(col);
    if (has2) row2PosToCol.push_back(col);
    cumulativeMatches[col] = cumulativeMatches[col - 1] + (has1 && has2 ? 1 : 0);
  }
}

// ---- minimal JSON reader (the reference vendors gason; only what RateModel::read needs) -----
namespace {
struct Json {
  enum Kind { Null, Num, Str, Arr, Obj } kind = Null;
  double num = 0;
  string str;
  vguard<Json> arr;
  vguard<std::pair<string, Json> > obj;
  const Json* find(const string& k) const {
    for (auto& kv : obj)
      if (kv.first == k) return &kv.second;
    return NULL;
  }
};

struct JsonParser {
  const string& s;
  size_t p;
  explicit JsonParser(const string& s) : s(s), p(0) {}
  void ws() { while (p < s.size() && isspace((unsigned char)s[p])) ++p; }
  Json parse() {
    ws();
    Require(p < s.size(), "JSON: unexpected end of input");
    Json j;
    const char c = s[p];
    if (c == '{') {
      j.kind = Json::Obj;
      ++p; ws();
      if (s[p] == '}') { ++p; return j; }
      while (true) {
        ws();
        Require(s[p] == '"', "JSON: expected string key at offset %u", (unsigned)p);
        Json k = parse();
        ws();
        Require(s[p] == ':', "JSON: expected ':' at offset %u", (unsigned)p);
        ++p;
        Json v = parse();
        j.obj.push_back(std::make_pair(k.str, v));
        ws();
        if (s[p] == ',') { ++p; continue; }
        Require(s[p] == '}', "JSON: expected '}' at offset %u", (unsigned)p);
        ++p;
        break;
      }
    } else if (c == '[') {
      j.kind = Json::Arr;
      ++p; ws();
      if (s[p] == ']') { ++p; return j; }
      while (true) {
        j.arr.push_back(parse());
        ws();
        if (s[p] == ',') { ++p; continue; }
        Require(s[p] == ']', "JSON: expected ']' at offset %u", (unsigned)p);
        ++p;
        break;
      }
    } else if (c == '"') {
      j.kind = Json::Str;
      ++p;
      while (p < s.size() && s[p] != '"') {
        if (s[p] == '\\' && p + 1 < s.size()) ++p;
        j.str.push_back(s[p++]);
      }
      ++p;
    } else if (c == '-' || isdigit((unsigned char)c)) {
      j.kind = Json::Num;
      char* end = NULL;
      j.num = strtod(s.c_str() + p, &end);
      p = end - s.c_str();
    } else {
      // true / false / null
      while (p < s.size() && isalpha((unsigned char)s[p])) ++p;
    }
    return j;
  }
};

double jnum(const Json& o, const char* key) {
  const Json* v = o.find(key);
  Assert(v != NULL && v->kind == Json::Num, "Couldn't find JSON number %s", key);
  return v->num;
}
}  // namespace

// ---- rate / probability models (reference src/model.cpp:172-232, 282-334, 374-391, 492-504) --
void RateModel::readFile(const char* filename) {
  std::ifstream in(filename);
  Require(in.good(), "Couldn't open %s", filename);
  std::stringstream ss;
  ss << in.rdbuf();
  read(ss.str());
}

void RateModel::read(const string& text) {
  Assert(subRate.empty(), "RateModel already initialized");
  JsonParser parser(text);
  const Json js = parser.parse();
  Assert(js.kind == Json::Obj, "JSON value is not an object");
  const Json* alph = js.find("alphabet");
  Assert(alph != NULL && alph->kind == Json::Str, "Couldn't find JSON tag alphabet");
  alphabet = alph->str;
  const Json* wc = js.find("wildcard");
  if (wc && wc->kind == Json::Str && !wc->str.empty()) wildcard = wc->str[0];
  insRate = jnum(js, "insrate");
  insExtProb = jnum(js, "insextprob");
  delRate = jnum(js, "delrate");
  delExtProb = jnum(js, "delextprob");
  auto readComponent = [&](const Json& jm) {
    const size_t A = alphabet.size();
    Mat sr(A, Vec(A, 0.));
    const Json* rm = jm.find("subrate");
    Assert(rm != NULL && rm->kind == Json::Obj, "Couldn't find JSON tag subrate");
    for (size_t i = 0; i < A; ++i) {
      const Json* row = rm->find(string(1, alphabet[i]));
      if (row)
        for (size_t j = 0; j < A; ++j)
          if (j != i) {
            const Json* r = row->find(string(1, alphabet[j]));
            if (r) {
              sr[i][j] += r->num;
              sr[i][i] -= r->num;
            }
          }
    }
    Vec ip;
    const Json* rp = jm.find("rootprob");
    if (rp) {
      ip.assign(A, 0.);
      for (size_t i = 0; i < A; ++i) {
        const Json* v = rp->find(string(1, alphabet[i]));
        if (v) ip[i] = v->num;
      }
    } else
      ip = getEqmProbVector(sr);
    const Json* w = jm.find("weight");
    cptWeight.push_back(w && w->kind == Json::Num ? w->num : 1);
    insProb.push_back(ip);
    subRate.push_back(sr);
  };
  const Json* mix = js.find("mixture");
  if (mix && mix->kind == Json::Arr)
    for (auto& c : mix->arr) readComponent(c);
  else
    readComponent(js);
  double norm = 0;
  for (double w : cptWeight) norm += w;
  for (auto& cw : cptWeight) cw /= norm;
}

// Least-squares solution of [R^T; 1 ... 1] pi = [0; 1] by Householder QR (the reference calls
// gsl_linalg_QR_decomp / QR_lssolve, src/model.cpp:282-303), then clamp at 0 and renormalise.
Vec RateModel::getEqmProbVector(const Mat& sr) {
  const size_t n = sr.size(), m = n + 1;
  vguard<Vec> a(m, Vec(n));
  for (size_t j = 0; j < n; ++j) {
    for (size_t i = 0; i < n; ++i) a[i][j] = sr[j][i];
    a[n][j] = 1;
  }
  Vec b(m, 0.);
  b[n] = 1;
  for (size_t k = 0; k < n; ++k) {
    double norm = 0;
    for (size_t i = k; i < m; ++i) norm += a[i][k] * a[i][k];
    norm = sqrt(norm);
    if (norm == 0) continue;
    const double alpha = a[k][k] > 0 ? -norm : norm;
    Vec v(m, 0.);
    for (size_t i = k; i < m; ++i) v[i] = a[i][k];
    v[k] -= alpha;
    double vnorm2 = 0;
    for (size_t i = k; i < m; ++i) vnorm2 += v[i] * v[i];
    if (vnorm2 == 0) continue;
    for (size_t j = k; j < n; ++j) {
      double dot = 0;
      for (size_t i = k; i < m; ++i) dot += v[i] * a[i][j];
      const double f = 2 * dot / vnorm2;
      for (size_t i = k; i < m; ++i) a[i][j] -= f * v[i];
    }
    double dot = 0;
    for (size_t i = k; i < m; ++i) dot += v[i] * b[i];
    const double f = 2 * dot / vnorm2;
    for (size_t i = k; i < m; ++i) b[i] -= f * v[i];
  }
  Vec eqm(n, 0.);
  for (size_t kk = n; kk-- > 0;) {
    double s = b[kk];
    for (size_t j = kk + 1; j < n; ++j) s -= a[kk][j] * eqm[j];
    eqm[kk] = a[kk][kk] != 0 ? s / a[kk][kk] : 0;
  }
  double eqmNorm = 0;
  for (size_t i = 0; i < n; ++i) {
    eqm[i] = std::max(0., eqm[i]);
    eqmNorm += eqm[i];
  }
  for (size_t i = 0; i < n; ++i) eqm[i] /= eqmNorm;
  return eqm;
}

// exp(R t) as the reference obtains it: gsl_linalg_exponential_ss(R t, GSL_PREC_DOUBLE), src/model.cpp:322-334.
// GSL is not vendored and not in this image, so its published algorithm is restated (Moler & Van Loan's method 3 as
// GSL's linalg/exponential.c tabulates it): divide by 2^j, a k-term Taylor series in Horner form, j squarings, (k, j)
// by the largest |element|; every multiply and add rounded separately, products accumulated in increasing inner
// index.  oracle/historian_oracle.py::sub_prob_matrix_ss performs the same operations in the same order.
static Mat matmul(const Mat& a, const Mat& b) {
  const size_t n = a.size();
  Mat c(n, Vec(n, 0.));
  for (size_t i = 0; i < n; ++i)
    for (size_t k = 0; k < n; ++k) {
      const double aik = a[i][k];
      if (aik != 0)
        for (size_t j = 0; j < n; ++j) c[i][j] += aik * b[k][j];
    }
  return c;
}

static void seriesShape(double supNorm, int& terms, int& squarings) {
  static const int table[6][2] = {{5, 1}, {5, 4}, {7, 5}, {9, 7}, {10, 10}, {8, 14}};
  static const double below[6] = {0.01, 0.1, 1., 10., 100., 1000.};
  for (int r = 0; r < 6; ++r)
    if (supNorm < below[r]) { terms = table[r][0]; squarings = table[r][1]; return; }
  terms = table[5][0];
  squarings = table[5][1] + (int)ceil(log(1.01 * supNorm / 1000.) / log(2.));
}

vguard<Mat> RateModel::getSubProbMatrix(double t) const {
  vguard<Mat> v;
  for (int c = 0; c < components(); ++c) {
    const size_t n = subRate[c].size();
    Mat b = subRate[c];
    double norm = 0;
    for (auto& row : b)
      for (auto& x : row) { x *= t; norm = std::max(norm, fabs(x)); }
    int terms, squarings;
    seriesShape(norm, terms, squarings);
    const double shrink = 1. / exp(log(2.) * squarings);
    for (auto& row : b)
      for (auto& x : row) x *= shrink;
    Mat eb = b;
    const double first = 1. / terms;
    for (auto& row : eb)
      for (auto& x : row) x *= first;
    for (size_t i = 0; i < n; ++i) eb[i][i] += 1.;
    for (int count = terms - 1; count >= 1; --count) {
      eb = matmul(b, eb);
      const double inv = 1. / count;
      for (auto& row : eb)
        for (auto& x : row) x *= inv;
      for (size_t i = 0; i < n; ++i) eb[i][i] += 1.;
    }
    for (int s = 0; s < squarings; ++s) eb = matmul(eb, eb);
    v.push_back(eb);
  }
  return v;
}

ProbModel::ProbModel(const RateModel& model, double t)
    : AlphabetOwner(model), t(t), ins(1 - exp(-model.insRate * t)), del(1 - exp(-model.delRate * t)),
      insExt(model.insExtProb), delExt(model.delExtProb), subMat(model.getSubProbMatrix(t)), insVec(model.insProb),
      cptWeight(model.cptWeight) {}

LogProbModel::LogProbModel(const ProbModel& pm) : logInsProb(pm.components()), logCptWeight(pm.components()) {
  for (int c = 0; c < pm.components(); ++c) {
    logCptWeight[c] = log(pm.cptWeight[c]);
    logInsProb[c] = log_vector(pm.insVec[c]);
  }
}

string pairParentName(const string& lChildName, double lTime, const string& rChildName, double rTime) {
  std::ostringstream o;
  o.unsetf(std::ios_base::floatfield);
  o << "(" << lChildName << ":" << lTime << "," << rChildName << ":" << rTime << ")";
  return o.str();
}

}  // namespace historian
