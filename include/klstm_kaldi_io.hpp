// include/klstm_kaldi_io.hpp -- Kaldi nnet1 text/binary stream format for the pieces the
// LstmProjectedStreams model file uses: tokens, int32/float basic types, float matrices and
// vectors.  Header-only, host-only C++ (no HIP, no Kaldi dependency).
//
// Format sources:
//   matrix write  /root/reference/google/matrix/kaldi-matrix.cc:1172-1212
//   matrix read   /root/reference/google/matrix/kaldi-matrix.cc:1243-1406  (FM, DM; text rows split
//                 by '\n' or ';', "[]" tolerated, inf/nan accepted)
//   token / basic-type / vector encoding: Kaldi base/io-funcs and matrix/kaldi-vector.cc are NOT
//   vendored in the reference; their well-known on-disk forms are restated here
//   ("<Tok> " ; binary ints/floats = one size byte + little-endian payload ; vector = "FV" + int32
//   dim + raw floats, text " [ a b c ]\n") and pinned by tests/test_component.py and tests/test_nnet.py against
//   hand-assembled byte strings and the reference's text samples (README.md:24-45,
//   google/feature_transform.nnet.txt).
#pragma once
#include <cctype>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <istream>
#include <limits>
#include <ostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace klstm_kaldi {

typedef int32_t int32;
typedef float BaseFloat;

// KALDI_ERR equivalent: throws std::runtime_error (kaldi-error.h behaviour the trainer relies on,
// bd-nnet-train-lstm-streams.cc:319-322).
struct ErrLine {
  std::ostringstream ss;
  template <class T> ErrLine &operator<<(const T &v) { ss << v; return *this; }
  [[noreturn]] void Throw() { throw std::runtime_error(ss.str()); }
};
#define KLSTM_ERR(msg) do { ::klstm_kaldi::ErrLine _e; _e << msg; _e.Throw(); } while (0)
#define KLSTM_ASSERT(cond) do { if (!(cond)) KLSTM_ERR("assertion failed: " #cond " (" << __FILE__ << ":" << __LINE__ << ")"); } while (0)

inline void WriteToken(std::ostream &os, bool /*binary*/, const char *token) {
  os << token << " ";
  if (os.fail()) KLSTM_ERR("model output stream failed while writing token '" << token << "'");
}
inline void WriteToken(std::ostream &os, bool binary, const std::string &t) { WriteToken(os, binary, t.c_str()); }

inline int Peek(std::istream &is, bool binary) {
  if (!binary) is >> std::ws;
  return is.peek();
}
inline void ReadToken(std::istream &is, bool binary, std::string *str) {
  if (!binary) is >> std::ws;
  is >> *str;
  if (is.fail()) KLSTM_ERR("model file: no token where one was expected (offset " << is.tellg() << ")");
  if (!isspace(is.peek())) KLSTM_ERR("model file: token '" << *str << "' is not followed by white space (next byte '" << (char)is.peek() << "')");
  is.get();  // the separator
}
inline void ExpectToken(std::istream &is, bool binary, const char *token) {
  std::string str;
  ReadToken(is, binary, &str);
  if (str != token) KLSTM_ERR("model file: found token '" << str << "' where '" << token << "' belongs");
}

template <class T> inline void WriteBasicType(std::ostream &os, bool binary, T t) {
  if (binary) {
    const char len = (char)sizeof(t);          // ints: +size, floats: size (both positive here)
    os.put(len);
    os.write(reinterpret_cast<const char *>(&t), sizeof(t));
  } else {
    os << t << " ";
  }
  if (os.fail()) KLSTM_ERR("model output stream failed while writing a scalar");
}
template <class T> inline void ReadBasicType(std::istream &is, bool binary, T *t) {
  if (binary) {
    const int len_c_in = is.get();
    if (len_c_in == -1) KLSTM_ERR("model file ends inside a binary scalar");
    const char len_c = (char)len_c_in;
    if (len_c != (char)sizeof(*t))
      KLSTM_ERR("model file: binary scalar is stored with " << (int)len_c << " bytes, this field has " << (int)sizeof(*t)
                << " (a model written with another integer / BaseFloat width?)");
    is.read(reinterpret_cast<char *>(t), sizeof(*t));
  } else {
    is >> *t;
  }
  if (is.fail()) KLSTM_ERR("model file: unreadable scalar near offset " << is.tellg());
}

// ---- matrices and vectors ------------------------------------------------------------------------
// On-disk forms (the format, not the code, is what has to match Kaldi):
//   binary matrix  "FM " int32 rows int32 cols, then rows*cols floats without row padding   ("DM " = doubles)
//   binary vector  "FV " int32 dim, then dim floats                                        ("DV " = doubles)
//   text matrix    " [" then per row "\n  v v v " and a closing "]\n"; an empty one is " [ ]\n"
//   text vector    " [ v v v ]\n"
// The text reader is a two-stage tokenizer of its own: (1) pull the bracketed body out of the stream as one string,
// (2) cut it into rows at '\n' / ';' and convert each blank-separated field with strtod (which already understands
// inf / infinity / nan in any case and maps out-of-range literals such as 1e+50 to +-inf after the float cast).
namespace detail {

[[noreturn]] inline void ReadFailure(const char *what, std::istream &is, std::streamoff start, const std::string &why) {
  KLSTM_ERR("cannot read " << what << " (" << why << "); the object started at stream offset " << start
            << ", the stream is now at " << is.tellg());
}

// Consumes "[ ... ]" plus the line end behind it and returns what stood between the brackets.  "[]" gives an empty body.
inline std::string TakeBracketedBody(std::istream &is, const char *what, std::streamoff start) {
  is >> std::ws;
  if (is.peek() != '[') {
    std::string got;
    is >> got;
    ReadFailure(what, is, start, got.empty() ? "the stream ended where '[' should be" : "found '" + got + "' where '[' should be");
  }
  is.get();
  std::string body;
  if (!std::getline(is, body, ']')) ReadFailure(what, is, start, "no closing ']' before the end of the stream");
  if (is.eof()) ReadFailure(what, is, start, "no closing ']' before the end of the stream");
  if (is.peek() == '\r') is.get();                   // the writer ends the object with a newline; eat one "\n" or "\r\n"
  if (is.peek() == '\n') is.get();
  return body;
}

// Converts the blank-separated fields of [p, end) and appends them to *out.
inline void ParseFields(const char *p, const char *end, std::vector<BaseFloat> *out, const char *what, std::istream &is,
                        std::streamoff start) {
  while (p < end) {
    if (isspace((unsigned char)*p)) { ++p; continue; }
    const char *q = p;
    while (q < end && !isspace((unsigned char)*q)) ++q;
    const std::string field(p, q);
    char *stop = nullptr;
    const double v = strtod(field.c_str(), &stop);
    if (stop == field.c_str() || *stop != '\0') ReadFailure(what, is, start, "'" + field + "' is not a number");
    out->push_back((BaseFloat)v);
    p = q;
  }
}

template <class Disk>
inline void ReadRaw(std::istream &is, size_t n, BaseFloat *dst) {
  if (n == 0) return;
  if (sizeof(Disk) == sizeof(BaseFloat)) { is.read(reinterpret_cast<char *>(dst), sizeof(BaseFloat) * n); return; }
  std::vector<Disk> tmp(n);
  is.read(reinterpret_cast<char *>(tmp.data()), sizeof(Disk) * n);
  for (size_t i = 0; i < n; i++) dst[i] = (BaseFloat)tmp[i];
}

}  // namespace detail

inline void WriteMatrix(std::ostream &os, bool binary, const BaseFloat *data, int32 rows, int32 cols, int32 stride) {
  if (binary) {
    WriteToken(os, binary, "FM");
    WriteBasicType(os, binary, rows);
    WriteBasicType(os, binary, cols);
    for (int32 r = 0; r < rows && cols > 0; r++)     // row by row: the file carries no row padding
      os.write(reinterpret_cast<const char *>(data + (size_t)r * stride), sizeof(BaseFloat) * (size_t)cols);
  } else if (cols == 0) {
    os << " [ ]\n";
  } else {
    std::ostringstream text;                         // assembled first: one write, and os keeps its own float formatting
    text.copyfmt(os);
    text << " [";
    for (int32 r = 0; r < rows; r++) {
      text << "\n  ";
      const BaseFloat *row = data + (size_t)r * stride;
      for (int32 c = 0; c < cols; c++) text << row[c] << ' ';
    }
    text << "]\n";
    os << text.str();
  }
  if (!os.good()) KLSTM_ERR("matrix of " << rows << " x " << cols << " could not be written: the output stream reports an error");
}

inline void ReadMatrix(std::istream &is, bool binary, std::vector<BaseFloat> *data, int32 *rows, int32 *cols) {
  const std::streamoff start = is.tellg();
  if (binary) {
    if (Peek(is, binary) == 'C') detail::ReadFailure("matrix", is, start, "compressed matrices (CM) are not supported");
    std::string kind;
    ReadToken(is, binary, &kind);
    if (kind != "FM" && kind != "DM") detail::ReadFailure("matrix", is, start, "type marker '" + kind + "' instead of FM / DM");
    int32 r = 0, c = 0;
    ReadBasicType(is, binary, &r);
    ReadBasicType(is, binary, &c);
    if (r < 0 || c < 0) detail::ReadFailure("matrix", is, start, "negative size");
    data->resize((size_t)r * c);
    if (kind == "FM") detail::ReadRaw<float>(is, data->size(), data->data());
    else detail::ReadRaw<double>(is, data->size(), data->data());
    if (is.fail()) detail::ReadFailure("matrix", is, start, "the stream ended inside the payload");
    *rows = r; *cols = c;
    return;
  }
  const std::string body = detail::TakeBracketedBody(is, "matrix", start);
  data->clear();
  int32 nrows = 0, ncols = 0;
  size_t a = 0;
  while (a <= body.size()) {                         // one row per '\n' / ';' separated piece; blank pieces are not rows
    size_t b = body.find_first_of("\n;", a);
    if (b == std::string::npos) b = body.size();
    const size_t before = data->size();
    detail::ParseFields(body.data() + a, body.data() + b, data, "matrix", is, start);
    const int32 got = (int32)(data->size() - before);
    if (got > 0) {
      if (nrows == 0) ncols = got;
      else if (got != ncols) {
        std::ostringstream why;
        why << "row " << nrows << " has " << got << " entries, the rows before it have " << ncols;
        detail::ReadFailure("matrix", is, start, why.str());
      }
      nrows++;
    }
    a = b + 1;
  }
  *rows = nrows; *cols = ncols;
}

// ---- vector ([UPSTREAM-unvendored] kaldi-vector.cc Write/Read: same framing with FV / DV and a single text row) ----
inline void WriteVector(std::ostream &os, bool binary, const BaseFloat *data, int32 dim) {
  if (binary) {
    WriteToken(os, binary, "FV");
    WriteBasicType(os, binary, dim);
    if (dim > 0) os.write(reinterpret_cast<const char *>(data), sizeof(BaseFloat) * (size_t)dim);
  } else {
    os << " [ ";
    for (int32 i = 0; i < dim; i++) os << data[i] << ' ';
    os << "]\n";
  }
  if (!os.good()) KLSTM_ERR("vector of " << dim << " elements could not be written: the output stream reports an error");
}

inline void ReadVector(std::istream &is, bool binary, std::vector<BaseFloat> *data) {
  const std::streamoff start = is.tellg();
  if (binary) {
    std::string kind;
    ReadToken(is, binary, &kind);
    if (kind != "FV" && kind != "DV") detail::ReadFailure("vector", is, start, "type marker '" + kind + "' instead of FV / DV");
    int32 dim = 0;
    ReadBasicType(is, binary, &dim);
    if (dim < 0) detail::ReadFailure("vector", is, start, "negative size");
    data->resize(dim);
    if (kind == "FV") detail::ReadRaw<float>(is, data->size(), data->data());
    else detail::ReadRaw<double>(is, data->size(), data->data());
    if (is.fail()) detail::ReadFailure("vector", is, start, "the stream ended inside the payload");
    return;
  }
  const std::string body = detail::TakeBracketedBody(is, "vector", start);
  if (body.find_first_of("\n;") != std::string::npos) {
    // a row separator inside the brackets: allowed only as trailing blank space (a matrix was probably meant)
    const size_t sep = body.find_first_of("\n;");
    if (body.find_first_not_of(" \t\r\n;", sep) != std::string::npos)
      detail::ReadFailure("vector", is, start, "more than one row between the brackets: this looks like a matrix");
  }
  data->clear();
  detail::ParseFields(body.data(), body.data() + body.size(), data, "vector", is, start);
}

// Kaldi files written through Output(..., binary=true, write_header=true) start with "\0B";
// text files have no header.  Returns the binary flag and leaves the stream after the header.
inline bool InitKaldiInputStream(std::istream &is) {
  if (is.peek() == '\0') {
    is.get();
    if (is.peek() != 'B') KLSTM_ERR("bad Kaldi binary header");
    is.get();
    return true;
  }
  return false;
}
inline void InitKaldiOutputStream(std::ostream &os, bool binary) {
  if (binary) { os.put('\0'); os.put('B'); }
}

}  // namespace klstm_kaldi
