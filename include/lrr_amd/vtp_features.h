// vtp_features.h — feature files without VTK.
//
// location_registration reads its two feature sets from VTK XML PolyData files
// (rrl/itkLocationRegistration.cxx:803-841: vtkXMLPolyDataReader on "<dir>_00.vtk") and converts them with
// vtkPolyDataToFeaturesICPFilter (cdcl/io/vtkPolyDataToFeaturesICPFilter.cxx:129-237): point coordinates
// plus the point-data arrays "errorProjectors" (9 floats), "strengths" (float), "shapes" (unsigned char),
// "normals" and "binormals" (3 floats each), as written by features/vtkITKPointSetToPolyDataFilter.cxx:147-250
// through vtkXMLPolyDataWriter.  VTK is not in this image, so read_vtp() parses that file format directly:
// <DataArray> with format "ascii", "binary" (inline base64) or "appended" (base64 or raw), header_type
// UInt32 / UInt64, little-endian, optionally zlib-compressed blocks (vtkZLibDataCompressor) — every
// combination vtkXMLPolyDataWriter can be configured to write.  features_from_vtp() is the filter's
// conversion, field by field.  Values are float on disk and widened to double, as in the reference.
#ifndef LRR_AMD_VTP_FEATURES_H_
#define LRR_AMD_VTP_FEATURES_H_

#include <zlib.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "lrr_amd/cdcl_types.h"

namespace lrr_amd {

struct vtp_arrays {
  size_t n = 0;
  std::vector<double> points;           // n x 3
  std::vector<double> errorProjectors;  // n x 9 (empty when the array is absent)
  std::vector<double> strengths;        // n
  std::vector<double> shapes;           // n
  std::vector<double> normals;          // n x 3
  std::vector<double> binormals;        // n x 3
};

namespace vtp_detail {

inline std::string attr(const std::string &tag, const std::string &name) {
  // value of name="..." inside one tag's text; empty when absent
  size_t p = 0;
  while ((p = tag.find(name, p)) != std::string::npos) {
    const bool start_ok = p == 0 || tag[p - 1] == ' ' || tag[p - 1] == '\t' || tag[p - 1] == '\n' || tag[p - 1] == '\r';
    size_t q = p + name.size();
    while (q < tag.size() && (tag[q] == ' ' || tag[q] == '\t')) ++q;
    if (start_ok && q < tag.size() && tag[q] == '=') {
      ++q;
      while (q < tag.size() && (tag[q] == ' ' || tag[q] == '\t')) ++q;
      if (q < tag.size() && (tag[q] == '"' || tag[q] == '\'')) {
        const char quote = tag[q];
        size_t e = tag.find(quote, q + 1);
        if (e == std::string::npos) return "";
        return tag.substr(q + 1, e - q - 1);
      }
    }
    p += name.size();
  }
  return "";
}

inline int b64_value(unsigned char c) {
  if (c >= 'A' && c <= 'Z') return c - 'A';
  if (c >= 'a' && c <= 'z') return c - 'a' + 26;
  if (c >= '0' && c <= '9') return c - '0' + 52;
  if (c == '+') return 62;
  if (c == '/') return 63;
  return -1;
}

// decodes base64 text starting at s[pos] until `want` bytes are produced, padding or a non-base64 character
// ends the unit; whitespace is skipped.  Returns the position after the consumed characters (a unit's
// padding included).
inline size_t b64_decode(const std::string &s, size_t pos, size_t want, std::vector<unsigned char> &out) {
  out.clear();
  unsigned acc = 0;
  int bits = 0;
  while (pos < s.size() && out.size() < want) {
    const unsigned char c = (unsigned char)s[pos];
    if (c == ' ' || c == '\n' || c == '\r' || c == '\t') { ++pos; continue; }
    if (c == '=') break;
    const int v = b64_value(c);
    if (v < 0) break;
    acc = (acc << 6) | (unsigned)v;
    bits += 6;
    ++pos;
    if (bits >= 8) {
      bits -= 8;
      out.push_back((unsigned char)((acc >> bits) & 0xffu));
    }
  }
  // finish the 4-character group the unit ends in (its last data characters and padding)
  while (pos < s.size() && s[pos] == '=') ++pos;
  return pos;
}

inline size_t type_size(const std::string &t) {
  if (t == "Float32" || t == "Int32" || t == "UInt32") return 4;
  if (t == "Float64" || t == "Int64" || t == "UInt64") return 8;
  if (t == "Int16" || t == "UInt16") return 2;
  if (t == "Int8" || t == "UInt8") return 1;
  throw std::runtime_error("vtp: unsupported DataArray type " + t);
}

inline double value_at(const unsigned char *p, const std::string &t) {
  if (t == "Float32") { float v; std::memcpy(&v, p, 4); return v; }
  if (t == "Float64") { double v; std::memcpy(&v, p, 8); return v; }
  if (t == "UInt8") return *p;
  if (t == "Int8") return (signed char)*p;
  if (t == "Int16") { int16_t v; std::memcpy(&v, p, 2); return v; }
  if (t == "UInt16") { uint16_t v; std::memcpy(&v, p, 2); return v; }
  if (t == "Int32") { int32_t v; std::memcpy(&v, p, 4); return v; }
  if (t == "UInt32") { uint32_t v; std::memcpy(&v, p, 4); return v; }
  if (t == "Int64") { int64_t v; std::memcpy(&v, p, 8); return (double)v; }
  if (t == "UInt64") { uint64_t v; std::memcpy(&v, p, 8); return (double)v; }
  throw std::runtime_error("vtp: unsupported DataArray type " + t);
}

inline uint64_t header_int(const unsigned char *p, size_t hsize) {
  if (hsize == 4) { uint32_t v; std::memcpy(&v, p, 4); return v; }
  uint64_t v; std::memcpy(&v, p, 8); return v;
}

struct file_info {
  size_t hsize = 4;       // header_type: UInt32 (VTK default) or UInt64
  bool compressed = false;
  bool appended_raw = false;
  size_t appended_begin = 0, appended_end = 0;  // the bytes after '_' of <AppendedData>
};

// the bytes of one binary data block (inline base64 text or a region of the appended section).  `limit`: the most bytes
// the caller can use (expected values x type size): every size word read from the file is checked against it and against
// what is left of the file BEFORE anything is sized or addressed with it (checks in subtraction form: no sum can wrap).
inline std::vector<unsigned char> block_bytes(const std::string &text, size_t pos, bool raw, const file_info &fi, uint64_t limit) {
  std::vector<unsigned char> out, tmp;
  const size_t h = fi.hsize;
  if (pos > text.size()) throw std::runtime_error("vtp: block offset beyond the file");
  const size_t left = text.size() - pos;  // bytes (or base64 characters) from pos to the end of the file
  if (!fi.compressed) {
    if (raw) {
      if (left < h) throw std::runtime_error("vtp: truncated raw block");
      const uint64_t nbytes = header_int((const unsigned char *)text.data() + pos, h);
      if (nbytes > left - h) throw std::runtime_error("vtp: truncated raw block");
      if (nbytes > limit) throw std::runtime_error("vtp: a block is larger than its array");
      out.assign(text.begin() + pos + h, text.begin() + pos + h + (size_t)nbytes);
      return out;
    }
    // base64: VTK encodes the length word as a unit of its own; other writers encode length + data together
    size_t p2 = b64_decode(text, pos, h, tmp);
    if (tmp.size() < h) throw std::runtime_error("vtp: truncated base64 header");
    const uint64_t nbytes = header_int(tmp.data(), h);
    if (nbytes > limit) throw std::runtime_error("vtp: a block is larger than its array");
    if (nbytes / 3 > left / 4 + 1) throw std::runtime_error("vtp: truncated base64 block");  // 4 characters carry 3 bytes
    // was the header a unit of its own (ended by padding) or the start of a joint unit?
    bool separate = p2 > 0 && text[p2 - 1] == '=';
    if (separate) {
      b64_decode(text, p2, (size_t)nbytes, out);
    } else {
      b64_decode(text, pos, h + (size_t)nbytes, tmp);
      if (tmp.size() < h) throw std::runtime_error("vtp: truncated base64 block");
      out.assign(tmp.begin() + h, tmp.end());
    }
    if (out.size() < nbytes) throw std::runtime_error("vtp: truncated base64 block");
    out.resize((size_t)nbytes);
    return out;
  }
  // compressed: [nblocks][block size][last block size][compressed size of each block], then the blocks
  std::vector<unsigned char> head;
  size_t data_pos;
  uint64_t nblocks, bsize, lsize;
  if (raw) {
    if (left < 3 * h) throw std::runtime_error("vtp: truncated compressed header");
    nblocks = header_int((const unsigned char *)text.data() + pos, h);
    if (nblocks > (left - 3 * h) / h) throw std::runtime_error("vtp: truncated compressed header");
    const size_t hbytes = (3 + (size_t)nblocks) * h;
    head.assign(text.begin() + pos, text.begin() + pos + hbytes);
    data_pos = pos + hbytes;
  } else {
    b64_decode(text, pos, 3 * h, tmp);
    if (tmp.size() < 3 * h) throw std::runtime_error("vtp: truncated compressed header");
    nblocks = header_int(tmp.data(), h);
    if (nblocks > left / h) throw std::runtime_error("vtp: truncated compressed header");  // (every size word takes more than h characters)
    const size_t hbytes = (3 + (size_t)nblocks) * h;
    data_pos = b64_decode(text, pos, hbytes, head);
    if (head.size() < hbytes) throw std::runtime_error("vtp: truncated compressed header");
  }
  bsize = header_int(head.data() + h, h);
  lsize = header_int(head.data() + 2 * h, h);
  // (bsize is the writer's nominal block size — 32 KB by default, whatever the array's size; what a block may inflate to is
  // checked block by block below)
  if (data_pos > text.size()) throw std::runtime_error("vtp: truncated compressed data");
  const uint64_t data_left = text.size() - data_pos;
  uint64_t ctotal = 0;
  for (uint64_t b = 0; b < nblocks; ++b) {
    const uint64_t csize = header_int(head.data() + (3 + b) * h, h);
    if (csize > data_left - ctotal) throw std::runtime_error("vtp: truncated compressed data");  // (ctotal <= data_left: no wrap)
    ctotal += csize;
  }
  std::vector<unsigned char> comp;
  if (raw) {
    comp.assign(text.begin() + data_pos, text.begin() + data_pos + (size_t)ctotal);
  } else {
    b64_decode(text, data_pos, (size_t)ctotal, comp);
    if (comp.size() < ctotal) throw std::runtime_error("vtp: truncated compressed data");
  }
  size_t cpos = 0;
  for (uint64_t b = 0; b < nblocks; ++b) {
    const uint64_t csize = header_int(head.data() + (3 + b) * h, h);
    const uint64_t usize = (b + 1 == nblocks && lsize != 0) ? lsize : bsize;
    if (csize > comp.size() - cpos) throw std::runtime_error("vtp: truncated compressed data");
    if (usize > limit - out.size()) throw std::runtime_error("vtp: the blocks inflate to more than their array");  // (out.size() <= limit)
    const size_t old = out.size();
    out.resize(old + (size_t)usize);
    uLongf got = (uLongf)usize;
    if (uncompress(out.data() + old, &got, comp.data() + cpos, (uLong)csize) != Z_OK || got != usize)
      throw std::runtime_error("vtp: zlib block does not inflate to its declared size");
    cpos += (size_t)csize;
  }
  return out;
}

// values of one <DataArray> element; tag = its opening tag, body_begin/body_end = its text content
inline std::vector<double> array_values(const std::string &text, const std::string &tag, size_t body_begin, size_t body_end,
                                        const file_info &fi, size_t expect) {
  const std::string type = attr(tag, "type");
  std::string format = attr(tag, "format");
  if (format.empty()) format = "ascii";
  std::vector<double> v;
  if (format == "ascii") {
    const char *p = text.data() + body_begin, *end = text.data() + body_end;
    v.reserve(std::min(expect, (body_end - body_begin) / 2 + 1));  // (a value takes at least two characters)
    while (p < end) {
      while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p;
      if (p >= end) break;
      char *q = nullptr;
      const double x = std::strtod(p, &q);
      if (q == p) throw std::runtime_error("vtp: bad number in an ascii DataArray");
      v.push_back(x);
      p = q;
    }
  } else {
    std::vector<unsigned char> bytes;
    const size_t ts = type_size(type);
    // (zlib deflates at most ~1030 : 1, everything else stores a value in at least one byte)
    if (expect / 1100 > text.size()) throw std::runtime_error("vtp: a DataArray declares more values than the file can hold");
    const uint64_t limit = (uint64_t)expect * ts;
    if (format == "binary") {
      bytes = block_bytes(text, body_begin, false, fi, limit);
    } else if (format == "appended") {
      const std::string off = attr(tag, "offset");
      if (off.empty()) throw std::runtime_error("vtp: appended DataArray without offset");
      const unsigned long long offv = std::strtoull(off.c_str(), nullptr, 10);
      if (fi.appended_end < fi.appended_begin || offv >= fi.appended_end - fi.appended_begin)
        throw std::runtime_error("vtp: appended offset beyond the data section");
      const size_t pos = fi.appended_begin + (size_t)offv;
      bytes = block_bytes(text, pos, fi.appended_raw, fi, limit);
    } else {
      throw std::runtime_error("vtp: unknown DataArray format " + format);
    }
    v.resize(bytes.size() / ts);
    for (size_t i = 0; i < v.size(); ++i) v[i] = value_at(bytes.data() + i * ts, type);
  }
  if (v.size() != expect) {
    std::ostringstream m;
    m << "vtp: DataArray \"" << attr(tag, "Name") << "\" holds " << v.size() << " values, expected " << expect;
    throw std::runtime_error(m.str());
  }
  return v;
}

}  // namespace vtp_detail

// Reads the first <Piece> of a VTK XML PolyData file.  Throws std::runtime_error with a message on anything
// it cannot interpret; arrays that are absent stay empty (the filter treats strengths / shapes as optional,
// cdcl/io/vtkPolyDataToFeaturesICPFilter.cxx:181-186).
inline void read_vtp(const std::string &path, vtp_arrays &out) {
  using namespace vtp_detail;
  std::ifstream f(path.c_str(), std::ios::binary);
  if (!f) throw std::runtime_error("vtp: cannot open " + path);
  std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  file_info fi;
  size_t p = text.find("<VTKFile");
  if (p == std::string::npos) throw std::runtime_error("vtp: not a VTK XML file: " + path);
  const size_t head_end = text.find('>', p);
  if (head_end == std::string::npos) throw std::runtime_error("vtp: unterminated <VTKFile> tag");
  const std::string head = text.substr(p, head_end - p);
  if (attr(head, "type") != "PolyData") throw std::runtime_error("vtp: VTKFile type is not PolyData");
  const std::string bo = attr(head, "byte_order");
  if (!bo.empty() && bo != "LittleEndian") throw std::runtime_error("vtp: only LittleEndian files are supported");
  const std::string ht = attr(head, "header_type");
  fi.hsize = ht == "UInt64" ? 8 : 4;
  fi.compressed = !attr(head, "compressor").empty();
  size_t xml_end = text.size();
  const size_t ap = text.find("<AppendedData");
  if (ap != std::string::npos) {
    const size_t tag_end = text.find('>', ap);
    if (tag_end == std::string::npos) throw std::runtime_error("vtp: unterminated <AppendedData> tag");
    const std::string atag = text.substr(ap, tag_end - ap);
    fi.appended_raw = attr(atag, "encoding") == "raw";
    size_t us = text.find('_', tag_end);
    if (us == std::string::npos) throw std::runtime_error("vtp: AppendedData without the leading underscore");
    fi.appended_begin = us + 1;
    const size_t close = text.rfind("</AppendedData>");
    fi.appended_end = close == std::string::npos ? text.size() : close;
    xml_end = ap;  // raw appended bytes may look like markup: never scan them for tags
  }
  const size_t piece = text.find("<Piece", p);
  if (piece == std::string::npos || piece > xml_end) throw std::runtime_error("vtp: no <Piece>");
  const size_t ptag_end = text.find('>', piece);
  if (ptag_end == std::string::npos || ptag_end > xml_end) throw std::runtime_error("vtp: unterminated <Piece> tag");
  const std::string ptag = text.substr(piece, ptag_end - piece);
  const std::string np = attr(ptag, "NumberOfPoints");
  if (np.empty()) throw std::runtime_error("vtp: <Piece> without NumberOfPoints");
  out = vtp_arrays();
  out.n = (size_t)std::strtoull(np.c_str(), nullptr, 10);
  if (out.n / 1100 > text.size()) throw std::runtime_error("vtp: NumberOfPoints exceeds what the file can hold");
  const size_t piece_end = std::min(text.find("</Piece>", piece), xml_end);
  const size_t pts_a = text.find("<Points", piece), pts_b = text.find("</Points>", piece);
  const size_t pd_a = text.find("<PointData", piece), pd_b = text.find("</PointData>", piece);
  size_t q = piece;
  bool have_points = false;
  while ((q = text.find("<DataArray", q)) != std::string::npos && q < piece_end) {
    const size_t tag_end = text.find('>', q);
    if (tag_end == std::string::npos) break;
    const std::string tag = text.substr(q, tag_end - q);
    const bool self_closed = tag_end > 0 && text[tag_end - 1] == '/';
    size_t body_begin = tag_end + 1, body_end = body_begin;
    if (!self_closed) {
      body_end = text.find("</DataArray>", body_begin);
      if (body_end == std::string::npos) throw std::runtime_error("vtp: unterminated <DataArray>");
    }
    const std::string ncs = attr(tag, "NumberOfComponents");
    const size_t nc = ncs.empty() ? 1 : (size_t)std::strtoull(ncs.c_str(), nullptr, 10);
    const bool in_points = pts_a != std::string::npos && q > pts_a && q < pts_b;
    const bool in_pd = pd_a != std::string::npos && q > pd_a && q < pd_b;
    const std::string name = attr(tag, "Name");
    if (in_points && !have_points) {
      if (nc != 3) throw std::runtime_error("vtp: <Points> array must have 3 components");
      out.points = array_values(text, tag, body_begin, body_end, fi, 3 * out.n);
      have_points = true;
    } else if (in_pd) {
      std::vector<double> *dst = nullptr;
      size_t want = 0;
      if (name == "errorProjectors") { dst = &out.errorProjectors; want = 9; }
      else if (name == "strengths") { dst = &out.strengths; want = 1; }
      else if (name == "shapes") { dst = &out.shapes; want = 1; }
      else if (name == "normals") { dst = &out.normals; want = 3; }
      else if (name == "binormals") { dst = &out.binormals; want = 3; }
      if (dst) {
        if (nc != want) throw std::runtime_error("vtp: array \"" + name + "\" has the wrong number of components");
        *dst = array_values(text, tag, body_begin, body_end, fi, want * out.n);
      }
    }
    q = self_closed ? tag_end + 1 : body_end;
  }
  if (!have_points && out.n > 0) throw std::runtime_error("vtp: no <Points> array");
}

// vtkPolyDataToFeaturesICPFilter::GenerateOutput (cdcl/io/vtkPolyDataToFeaturesICPFilter.cxx:129-237):
// location, error projector (row-major 3 x 3), strength (0 when absent), shape (0 CORNER, 1 TUBE, 2 SHEET) and
// directions_: CORNER {normal x binormal, binormal, normal}, TUBE {binormal, normal}, SHEET {normal}
template <class FeatureT>
inline std::vector<typename FeatureT::sptr> features_from_vtp(const vtp_arrays &a) {
  typedef vnl_vector_fixed<double, 3> vec3;
  if (a.errorProjectors.size() != 9 * a.n) throw std::runtime_error("vtp: the errorProjectors array is required");
  if (a.normals.size() != 3 * a.n || a.binormals.size() != 3 * a.n) throw std::runtime_error("vtp: the normals and binormals arrays are required");
  std::vector<typename FeatureT::sptr> out;
  out.reserve(a.n);
  for (size_t i = 0; i < a.n; ++i) {
    FeatureT *f = new FeatureT;
    for (int d = 0; d < 3; ++d) f->location_[d] = a.points[3 * i + d];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) f->error_projector_(r, c) = a.errorProjectors[9 * i + 3 * r + c];
    f->strength_ = (decltype(f->strength_))(a.strengths.empty() ? 0.0 : a.strengths[i]);
    const unsigned char shape = (unsigned char)(a.shapes.empty() ? 0.0 : a.shapes[i]);
    const vec3 normal(a.normals[3 * i], a.normals[3 * i + 1], a.normals[3 * i + 2]);
    const vec3 binormal(a.binormals[3 * i], a.binormals[3 * i + 1], a.binormals[3 * i + 2]);
    switch (shape) {
      case 0: {
        f->shape_ = FeatureT::CORNER;
        const vec3 tangent(normal[1] * binormal[2] - normal[2] * binormal[1], normal[2] * binormal[0] - normal[0] * binormal[2],
                           normal[0] * binormal[1] - normal[1] * binormal[0]);  // vnl_cross_3d(normal, binormal)
        f->directions_.push_back(tangent);
        f->directions_.push_back(binormal);
        f->directions_.push_back(normal);
        break;
      }
      case 1:
        f->shape_ = FeatureT::TUBE;
        f->directions_.push_back(binormal);
        f->directions_.push_back(normal);
        break;
      case 2:
        f->shape_ = FeatureT::SHEET;
        f->directions_.push_back(normal);
        break;
      default:  // the reference leaves `shape` uninitialised here: treated as a corner without directions
        f->shape_ = FeatureT::CORNER;
        break;
    }
    out.push_back(f);
  }
  return out;
}

}  // namespace lrr_amd
#endif
