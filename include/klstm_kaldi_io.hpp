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
  if (os.fail()) KLSTM_ERR("WriteToken: stream failure");
}
inline void WriteToken(std::ostream &os, bool binary, const std::string &t) { WriteToken(os, binary, t.c_str()); }

inline int Peek(std::istream &is, bool binary) {
  if (!binary) is >> std::ws;
  return is.peek();
}
inline void ReadToken(std::istream &is, bool binary, std::string *str) {
  if (!binary) is >> std::ws;
  is >> *str;
  if (is.fail()) KLSTM_ERR("ReadToken, failed to read token at file position " << is.tellg());
  if (!isspace(is.peek())) KLSTM_ERR("ReadToken, expected space after token, saw instead " << (char)is.peek());
  is.get();  // consume the space
}
inline void ExpectToken(std::istream &is, bool binary, const char *token) {
  std::string str;
  ReadToken(is, binary, &str);
  if (str != token) KLSTM_ERR("Expected token \"" << token << "\", got instead \"" << str << "\".");
}

template <class T> inline void WriteBasicType(std::ostream &os, bool binary, T t) {
  if (binary) {
    const char len = (char)sizeof(t);          // ints: +size, floats: size (both positive here)
    os.put(len);
    os.write(reinterpret_cast<const char *>(&t), sizeof(t));
  } else {
    os << t << " ";
  }
  if (os.fail()) KLSTM_ERR("WriteBasicType: stream failure");
}
template <class T> inline void ReadBasicType(std::istream &is, bool binary, T *t) {
  if (binary) {
    const int len_c_in = is.get();
    if (len_c_in == -1) KLSTM_ERR("ReadBasicType: encountered end of stream.");
    const char len_c = (char)len_c_in;
    if (len_c != (char)sizeof(*t))
      KLSTM_ERR("ReadBasicType: did not get expected integer type, " << (int)len_c << " vs. " << (int)sizeof(*t)
                << ".  You can change this code to successfully read it later, if needed.");
    is.read(reinterpret_cast<char *>(t), sizeof(*t));
  } else {
    is >> *t;
  }
  if (is.fail()) KLSTM_ERR("Read failure in ReadBasicType, file position is " << is.tellg());
}

// ---- matrix ------------------------------------------------------------------------------------
inline void WriteMatrix(std::ostream &os, bool binary, const BaseFloat *data, int32 rows, int32 cols, int32 stride) {
  if (!os.good()) KLSTM_ERR("Failed to write matrix to stream: stream not good");
  if (binary) {
    WriteToken(os, binary, "FM");
    WriteBasicType(os, binary, rows);
    WriteBasicType(os, binary, cols);
    if (stride == cols) os.write(reinterpret_cast<const char *>(data), sizeof(BaseFloat) * (size_t)rows * cols);
    else for (int32 i = 0; i < rows; i++) os.write(reinterpret_cast<const char *>(data + (size_t)i * stride), sizeof(BaseFloat) * cols);
    if (!os.good()) KLSTM_ERR("Failed to write matrix to stream");
  } else {
    if (cols == 0) { os << " [ ]\n"; return; }
    os << " [";
    for (int32 i = 0; i < rows; i++) {
      os << "\n  ";
      for (int32 j = 0; j < cols; j++) os << data[(size_t)i * stride + j] << " ";
    }
    os << "]\n";
  }
}

inline bool ParseTextNumber(std::istream &is, BaseFloat *out, std::string *err) {
  // a number, or inf / nan in any case (kaldi-matrix.cc:1366-1392)
  const int i = is.peek();
  if ((i >= '0' && i <= '9') || i == '-' || i == '+' || i == '.') {
    // operator>> into float fails on overflow-to-inf tokens such as "1e+50"; go through double
    double d;
    is >> d;
    if (is.fail()) { *err = "Stream failure/EOF while reading matrix data."; return false; }
    *out = (BaseFloat)d;
    return true;
  }
  std::string str;
  is >> str;
  std::string low;
  for (char c : str) low.push_back((char)tolower(c));
  if (low == "inf" || low == "infinity") { *out = std::numeric_limits<BaseFloat>::infinity(); return true; }
  if (low == "-inf" || low == "-infinity") { *out = -std::numeric_limits<BaseFloat>::infinity(); return true; }
  if (low == "nan") { *out = std::numeric_limits<BaseFloat>::quiet_NaN(); return true; }
  *err = "Expecting numeric matrix data, got " + str;
  return false;
}

inline void ReadMatrix(std::istream &is, bool binary, std::vector<BaseFloat> *data, int32 *rows, int32 *cols) {
  std::ostringstream specific_error;
  const std::streamoff pos_at_start = is.tellg();
  if (binary) {
    const int peekval = Peek(is, binary);
    if (peekval == 'C') KLSTM_ERR("Failed to read matrix from stream: CompressedMatrix (CM) is not supported by this reader");
    std::string token;
    ReadToken(is, binary, &token);
    if (token != "FM" && token != "DM") {
      specific_error << ": Expected token FM, got " << token;
      goto bad;
    }
    {
      int32 r, c;
      ReadBasicType(is, binary, &r);
      ReadBasicType(is, binary, &c);
      if (r < 0 || c < 0) { specific_error << ": negative dimensions"; goto bad; }
      data->resize((size_t)r * c);
      if (token == "FM") {
        if (r * c != 0) is.read(reinterpret_cast<char *>(data->data()), sizeof(BaseFloat) * (size_t)r * c);
      } else {   // DM: double on disk, converted (kaldi-matrix.cc:1276-1283)
        std::vector<double> tmp((size_t)r * c);
        if (r * c != 0) is.read(reinterpret_cast<char *>(tmp.data()), sizeof(double) * (size_t)r * c);
        for (size_t i = 0; i < tmp.size(); i++) (*data)[i] = (BaseFloat)tmp[i];
      }
      if (is.fail()) goto bad;
      *rows = r; *cols = c;
      return;
    }
  } else {
    std::string str;
    is >> str;
    if (is.fail()) { specific_error << ": Expected \"[\", got EOF"; goto bad; }
    if (str == "[]") { data->clear(); *rows = 0; *cols = 0; return; }
    if (str != "[") { specific_error << ": Expected \"[\", got \"" << str << '"'; goto bad; }
    std::vector<std::vector<BaseFloat> > rowsv;
    std::vector<BaseFloat> cur;
    while (1) {
      const int i = is.peek();
      if (i == -1) { specific_error << "Got EOF while reading matrix data"; goto bad; }
      if ((char)i == ']') {
        is.get();
        const int j = is.peek();
        if ((char)j == '\r') { is.get(); is.get(); }
        else if ((char)j == '\n') { is.get(); }
        if (!cur.empty()) rowsv.push_back(cur);
        if (rowsv.empty()) { data->clear(); *rows = 0; *cols = 0; return; }
        const size_t nc = rowsv[0].size();
        data->resize(rowsv.size() * nc);
        for (size_t r = 0; r < rowsv.size(); r++) {
          if (rowsv[r].size() != nc) {
            specific_error << "Matrix has inconsistent #cols: " << nc << " vs." << rowsv[r].size() << " (processing row" << r << ")";
            goto bad;
          }
          memcpy(data->data() + r * nc, rowsv[r].data(), nc * sizeof(BaseFloat));
        }
        *rows = (int32)rowsv.size(); *cols = (int32)nc;
        return;
      } else if ((char)i == '\n' || (char)i == ';') {
        is.get();
        if (!cur.empty()) { rowsv.push_back(cur); cur.clear(); }
      } else if (isspace(i)) {
        is.get();
      } else {
        BaseFloat v;
        std::string err;
        if (!ParseTextNumber(is, &v, &err)) { specific_error << err; goto bad; }
        cur.push_back(v);
      }
    }
  }
bad:
  KLSTM_ERR("Failed to read matrix from stream.  " << specific_error.str() << " File position at start is "
            << pos_at_start << ", currently " << is.tellg());
}

// ---- vector ([UPSTREAM-unvendored] kaldi-vector.cc Write/Read) -----------------------------------
inline void WriteVector(std::ostream &os, bool binary, const BaseFloat *data, int32 dim) {
  if (!os.good()) KLSTM_ERR("Failed to write vector to stream: stream not good");
  if (binary) {
    WriteToken(os, binary, "FV");
    WriteBasicType(os, binary, dim);
    os.write(reinterpret_cast<const char *>(data), sizeof(BaseFloat) * (size_t)dim);
  } else {
    os << " [ ";
    for (int32 i = 0; i < dim; i++) os << data[i] << " ";
    os << "]\n";
  }
  if (!os.good()) KLSTM_ERR("Failed to write vector to stream");
}

inline void ReadVector(std::istream &is, bool binary, std::vector<BaseFloat> *data) {
  std::ostringstream specific_error;
  const std::streamoff pos_at_start = is.tellg();
  if (binary) {
    std::string token;
    ReadToken(is, binary, &token);
    if (token != "FV" && token != "DV") { specific_error << ": Expected token FV, got " << token; goto bad; }
    int32 dim;
    ReadBasicType(is, binary, &dim);
    if (dim < 0) { specific_error << ": negative dimension"; goto bad; }
    data->resize(dim);
    if (token == "FV") { if (dim) is.read(reinterpret_cast<char *>(data->data()), sizeof(BaseFloat) * (size_t)dim); }
    else {
      std::vector<double> tmp(dim);
      if (dim) is.read(reinterpret_cast<char *>(tmp.data()), sizeof(double) * (size_t)dim);
      for (int32 i = 0; i < dim; i++) (*data)[i] = (BaseFloat)tmp[i];
    }
    if (is.fail()) { specific_error << ": Error reading vector data (binary mode); truncated stream?"; goto bad; }
    return;
  } else {
    std::string s;
    is >> s;
    if (is.fail()) { specific_error << "EOF while trying to read vector."; goto bad; }
    if (s == "[]") { data->clear(); return; }
    if (s != "[") { specific_error << "Expected \"[\" but got " << s; goto bad; }
    data->clear();
    while (1) {
      const int i = is.peek();
      if (i == -1) { specific_error << "EOF while reading vector data."; goto bad; }
      if ((char)i == ']') {
        is.get();
        const int j = is.peek();
        if ((char)j == '\r') { is.get(); is.get(); }
        else if ((char)j == '\n') { is.get(); }
        return;
      }
      if ((char)i == '\n' || (char)i == ';') { specific_error << "Newline found while reading vector (maybe it's a matrix?)"; goto bad; }
      if (isspace(i)) { is.get(); continue; }
      BaseFloat v;
      std::string err;
      if (!ParseTextNumber(is, &v, &err)) { specific_error << err; goto bad; }
      data->push_back(v);
    }
  }
bad:
  KLSTM_ERR("Failed to read vector from stream.  " << specific_error.str() << " File position at start is "
            << pos_at_start << ", currently " << is.tellg());
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
