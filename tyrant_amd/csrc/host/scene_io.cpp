// scene_io.cpp -- the import half of Scene::Load for PLY files and image export of the resolved frame
// (SURVEY.md section 8f-2).
//
// Reference: Scene::Load reads a file through assimp with aiProcess_Triangulate (Scene.cpp:4-5), keeps the
// FIRST mesh only (static_mesh.cpp:6), copies its vertices with y and z exchanged (static_mesh.cpp:17), undoes
// that exchange (Scene.cpp:10) -- the two cancel -- and emits one Triangle{vert, e1 = v1 - v0, e2 = v2 - v0}
// per face (Scene.cpp:39-45).  assimp is not available here (Windows .lib only in the reference tree), so this
// is a PLY reader with the same conventions: polygons are triangulated as fans, extra vertex properties
// (normals, uv) are skipped, `{ ... }` annotations in the header and trailing text after the face list -- both
// present in the reference's own Data/cube.ply -- are tolerated.
//
// Export replaces the GL blit (interop.cpp:50): the RGBA32F buffer written by tyr_resolve goes to a binary PPM
// (8-bit, already tonemapped by kernel.cu:661's Reinhard + gamma) or a PFM (raw float, bottom-up rows).
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "host.hpp"

namespace {

enum class PlyFormat { Ascii, BinaryLE };

struct PlyProperty {
	std::string name;
	std::string type;      // scalar type, or the item type of a list
	std::string countType; // non-empty for list properties
};
struct PlyElement {
	std::string name;
	size_t count = 0;
	std::vector<PlyProperty> props;
};

size_t type_size(const std::string& t) {
	if (t == "char" || t == "uchar" || t == "int8" || t == "uint8")
		return 1;
	if (t == "short" || t == "ushort" || t == "int16" || t == "uint16")
		return 2;
	if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32")
		return 4;
	if (t == "double" || t == "float64")
		return 8;
	return 0;
}

// one binary little-endian scalar -> double
bool read_binary(std::istream& in, const std::string& t, double& out) {
	unsigned char b[8] = {};
	const size_t n = type_size(t);
	if (n == 0 || !in.read(reinterpret_cast<char*>(b), static_cast<std::streamsize>(n)))
		return false;
	if (t == "char" || t == "int8")
		out = static_cast<signed char>(b[0]);
	else if (t == "uchar" || t == "uint8")
		out = b[0];
	else if (t == "short" || t == "int16") {
		int16_t v;
		std::memcpy(&v, b, 2);
		out = v;
	} else if (t == "ushort" || t == "uint16") {
		uint16_t v;
		std::memcpy(&v, b, 2);
		out = v;
	} else if (t == "int" || t == "int32") {
		int32_t v;
		std::memcpy(&v, b, 4);
		out = v;
	} else if (t == "uint" || t == "uint32") {
		uint32_t v;
		std::memcpy(&v, b, 4);
		out = v;
	} else if (t == "float" || t == "float32") {
		float v;
		std::memcpy(&v, b, 4);
		out = v;
	} else {
		double v;
		std::memcpy(&v, b, 8);
		out = v;
	}
	return true;
}

std::string strip_annotation(std::string line) {
	const size_t brace = line.find('{');
	if (brace != std::string::npos)
		line.erase(brace);
	while (!line.empty() && std::isspace(static_cast<unsigned char>(line.back())))
		line.pop_back();
	return line;
}

} // namespace

extern "C" {

void tyr_free(void* p) { std::free(p); }

} // extern "C"

namespace {

int load_ply(const char* path, tyr_triangle** prims_out) {
	std::ifstream in(path, std::ios::binary);
	if (!in)
		return TYR_ERR_INVALID;
	// every count in the header is checked against this before anything is sized from it
	in.seekg(0, std::ios::end);
	const uint64_t fileSize = static_cast<uint64_t>(std::max<std::streamoff>(in.tellg(), 0));
	in.seekg(0, std::ios::beg);
	std::string line;
	if (!std::getline(in, line) || strip_annotation(line).rfind("ply", 0) != 0)
		return TYR_ERR_INVALID;
	PlyFormat fmt = PlyFormat::Ascii;
	std::vector<PlyElement> elements;
	bool headerDone = false;
	while (std::getline(in, line)) {
		line = strip_annotation(line);
		if (line.empty())
			continue;
		std::istringstream ls(line);
		std::string kw;
		ls >> kw;
		if (kw == "format") {
			std::string f;
			ls >> f;
			if (f == "ascii")
				fmt = PlyFormat::Ascii;
			else if (f == "binary_little_endian")
				fmt = PlyFormat::BinaryLE;
			else
				return TYR_ERR_UNSUPPORTED;
		} else if (kw == "element") {
			PlyElement e;
			ls >> e.name >> e.count;
			// an element instance occupies at least one byte in either format: a larger count is a corrupt (or hostile) header
			if (!ls || e.count > fileSize)
				return TYR_ERR_INVALID;
			elements.push_back(e);
		} else if (kw == "property" && !elements.empty()) {
			PlyProperty p;
			std::string t;
			ls >> t;
			if (t == "list") {
				ls >> p.countType >> p.type >> p.name;
			} else {
				p.type = t;
				ls >> p.name;
			}
			if (type_size(p.type) == 0 || (!p.countType.empty() && type_size(p.countType) == 0))
				return TYR_ERR_UNSUPPORTED;
			elements.back().props.push_back(p);
		} else if (kw == "end_header") {
			headerDone = true;
			break;
		} // comment, obj_info: ignored
	}
	if (!headerDone)
		return TYR_ERR_INVALID;

	std::vector<float> verts; // x y z per vertex
	std::vector<tyr_triangle> tris;
	// ASCII values: whitespace separated; `{ ... }` annotations may follow values on a data line too
	// (the reference's Data/cube.ply: "0 0 0    { start of vertex list }")
	auto next_ascii = [&](double& v) -> bool {
		for (;;) {
			int c = in.peek();
			while (c != EOF && std::isspace(c)) {
				in.get();
				c = in.peek();
			}
			if (c == EOF)
				return false;
			if (c == '{') {
				while (c != EOF && c != '}')
					c = in.get();
				continue;
			}
			break;
		}
		std::string tok;
		for (int c = in.peek(); c != EOF && !std::isspace(c) && c != '{'; c = in.peek())
			tok.push_back(static_cast<char>(in.get()));
		char* end = nullptr;
		v = std::strtod(tok.c_str(), &end);
		return end != tok.c_str() && *end == '\0';
	};
	auto next_value = [&](const std::string& type, double& v) -> bool {
		if (fmt == PlyFormat::Ascii)
			return next_ascii(v);
		return read_binary(in, type, v);
	};
	for (const PlyElement& e : elements) {
		if (e.name == "vertex") {
			int ix = -1, iy = -1, iz = -1;
			for (size_t k = 0; k < e.props.size(); ++k) {
				if (!e.props[k].countType.empty())
					return TYR_ERR_UNSUPPORTED;
				if (e.props[k].name == "x")
					ix = static_cast<int>(k);
				else if (e.props[k].name == "y")
					iy = static_cast<int>(k);
				else if (e.props[k].name == "z")
					iz = static_cast<int>(k);
			}
			if (ix < 0 || iy < 0 || iz < 0)
				return TYR_ERR_INVALID;
			verts.resize(e.count * 3);
			for (size_t i = 0; i < e.count; ++i) {
				for (size_t k = 0; k < e.props.size(); ++k) {
					double v;
					if (!next_value(e.props[k].type, v))
						return TYR_ERR_INVALID;
					const bool coord = static_cast<int>(k) == ix || static_cast<int>(k) == iy || static_cast<int>(k) == iz;
					if (coord && !std::isfinite(v))
						return TYR_ERR_INVALID; // non-finite geometry is rejected at upload anyway: say so here
					if (static_cast<int>(k) == ix)
						verts[3 * i + 0] = static_cast<float>(v);
					else if (static_cast<int>(k) == iy)
						verts[3 * i + 1] = static_cast<float>(v);
					else if (static_cast<int>(k) == iz)
						verts[3 * i + 2] = static_cast<float>(v);
				}
			}
		} else if (e.name == "face") {
			const size_t nVerts = verts.size() / 3;
			tris.reserve(e.count);
			std::vector<uint32_t> idx;
			for (size_t i = 0; i < e.count; ++i) {
				for (const PlyProperty& p : e.props) {
					double v;
					if (p.countType.empty()) {
						if (!next_value(p.type, v))
							return TYR_ERR_INVALID;
						continue;
					}
					if (!next_value(p.countType, v) || !(v >= 0 && v <= 255)) // (also false for NaN)
						return TYR_ERR_INVALID;
					const size_t n = static_cast<size_t>(v);
					const bool isIndexList = (p.name == "vertex_indices" || p.name == "vertex_index");
					idx.clear();
					for (size_t k = 0; k < n; ++k) {
						if (!next_value(p.type, v))
							return TYR_ERR_INVALID;
						if (isIndexList) {
							if (!(v >= 0 && v < static_cast<double>(nVerts))) // NaN, negative and out-of-range indices end here, before the cast
								return TYR_ERR_INVALID;
							idx.push_back(static_cast<uint32_t>(v));
						}
					}
					// aiProcess_Triangulate: a convex polygon becomes a fan around its first vertex
					for (size_t k = 1; isIndexList && k + 1 < idx.size(); ++k) {
						const float* a = &verts[3 * idx[0]];
						const float* b = &verts[3 * idx[k]];
						const float* c = &verts[3 * idx[k + 1]];
						tyr_triangle t{};
						for (int d = 0; d < 3; ++d) { // Scene.cpp:39-45
							t.vert[d] = a[d];
							t.e1[d] = b[d] - a[d];
							t.e2[d] = c[d] - a[d];
						}
						tris.push_back(t);
					}
				}
			}
		} else {
			// other elements (edge, material, ...): must still be consumed in order
			for (size_t i = 0; i < e.count; ++i) {
				for (const PlyProperty& p : e.props) {
					double v;
					if (p.countType.empty()) {
						if (!next_value(p.type, v))
							return TYR_ERR_INVALID;
					} else {
						if (!next_value(p.countType, v) || !(v >= 0 && v <= static_cast<double>(fileSize)))
							return TYR_ERR_INVALID;
						for (size_t k = 0, n = static_cast<size_t>(v); k < n; ++k)
							if (!next_value(p.type, v))
								return TYR_ERR_INVALID;
					}
				}
			}
		}
	}
	if (tris.size() > static_cast<size_t>(INT32_MAX))
		return TYR_ERR_UNSUPPORTED;
	if (!tris.empty()) {
		*prims_out = static_cast<tyr_triangle*>(std::malloc(tris.size() * sizeof(tyr_triangle)));
		if (!*prims_out)
			return TYR_ERR_OOM;
		std::memcpy(*prims_out, tris.data(), tris.size() * sizeof(tyr_triangle));
	}
	return static_cast<int>(tris.size());
}

// fwrite / fclose results folded into one status: a short write (full disk, closed pipe) is an error, not TYR_OK
struct OutFile {
	FILE* fp;
	bool ok = true;
	explicit OutFile(const char* path) : fp(std::fopen(path, "wb")) {}
	void write(const void* p, size_t size, size_t n) { ok = ok && fp && std::fwrite(p, size, n, fp) == n; }
	int close() {
		if (!fp)
			return TYR_ERR_IO;
		ok = (std::fclose(fp) == 0) && ok;
		fp = nullptr;
		return ok ? TYR_OK : TYR_ERR_IO;
	}
	~OutFile() {
		if (fp)
			std::fclose(fp);
	}
};

// nothing may leave an extern "C" function by exception (the caller may be C, or Python through ctypes)
template <class F>
int guarded(F&& f) {
	try {
		return f();
	} catch (const std::bad_alloc&) {
		return TYR_ERR_OOM;
	} catch (const std::exception&) {
		return TYR_ERR_INVALID;
	}
}

} // namespace

extern "C" {

int tyr_load_ply(const char* path, tyr_triangle** prims_out) {
	if (!path || !prims_out)
		return TYR_ERR_INVALID;
	*prims_out = nullptr;
	const int n = guarded([&] { return load_ply(path, prims_out); });
	if (n < 0 && *prims_out) {
		std::free(*prims_out);
		*prims_out = nullptr;
	}
	return n;
}

int tyr_write_ppm(const char* path, const float* rgba, uint32_t width, uint32_t height) {
	if (!path || !rgba || width == 0 || height == 0)
		return TYR_ERR_INVALID;
	return guarded([&]() -> int {
	OutFile out(path);
	if (!out.fp)
		return TYR_ERR_IO;
	char head[64];
	const int hn = std::snprintf(head, sizeof head, "P6 %u %u 255\n", width, height);
	out.write(head, 1, static_cast<size_t>(hn));
	std::vector<unsigned char> row(static_cast<size_t>(width) * 3);
	for (uint32_t y = 0; y < height; ++y) {
		for (uint32_t x = 0; x < width; ++x) {
			const float* p = &rgba[4 * (static_cast<size_t>(y) * width + x)];
			for (int c = 0; c < 3; ++c) {
				float v = p[c];
				v = (v != v) ? 0.0f : std::min(std::max(v, 0.0f), 1.0f); // pixels without a completed path resolve to NaN (0/0, kernel.cu:658)
				row[3 * x + c] = static_cast<unsigned char>(v * 255.0f + 0.5f);
			}
		}
		out.write(row.data(), 1, row.size());
	}
	return out.close();
	});
}

// PNG, 8-bit RGB, "stored" deflate blocks (no compression: the file is W*H*3 bytes plus a few per row block), so that the
// resolved frame opens in any viewer without a dependency on zlib
namespace {
struct Crc32 {
	uint32_t table[256];
	Crc32() {
		for (uint32_t n = 0; n < 256; ++n) {
			uint32_t c = n;
			for (int k = 0; k < 8; ++k)
				c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
			table[n] = c;
		}
	}
	uint32_t run(uint32_t crc, const unsigned char* p, size_t n) const {
		for (size_t i = 0; i < n; ++i)
			crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
		return crc;
	}
};
void put_be32(std::vector<unsigned char>& v, uint32_t x) {
	v.push_back(static_cast<unsigned char>(x >> 24));
	v.push_back(static_cast<unsigned char>(x >> 16));
	v.push_back(static_cast<unsigned char>(x >> 8));
	v.push_back(static_cast<unsigned char>(x));
}
void write_chunk(OutFile& out, const Crc32& crc, const char type[4], const std::vector<unsigned char>& data) {
	std::vector<unsigned char> head;
	put_be32(head, static_cast<uint32_t>(data.size()));
	out.write(head.data(), 1, 4);
	out.write(type, 1, 4);
	if (!data.empty())
		out.write(data.data(), 1, data.size());
	uint32_t c = crc.run(0xFFFFFFFFu, reinterpret_cast<const unsigned char*>(type), 4);
	c = crc.run(c, data.data(), data.size()) ^ 0xFFFFFFFFu;
	std::vector<unsigned char> tail;
	put_be32(tail, c);
	out.write(tail.data(), 1, 4);
}
} // namespace

int tyr_write_png(const char* path, const float* rgba, uint32_t width, uint32_t height) {
	if (!path || !rgba || width == 0 || height == 0 || static_cast<uint64_t>(width) * height > (1ull << 28))
		return TYR_ERR_INVALID;
	return guarded([&]() -> int {
	OutFile out(path);
	if (!out.fp)
		return TYR_ERR_IO;
	static const Crc32 crc;
	static const unsigned char magic[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
	out.write(magic, 1, 8);
	std::vector<unsigned char> ihdr;
	put_be32(ihdr, width);
	put_be32(ihdr, height);
	const unsigned char rest[5] = { 8, 2, 0, 0, 0 }; // 8 bits, colour type 2 (RGB), deflate, adaptive filtering, no interlace
	ihdr.insert(ihdr.end(), rest, rest + 5);
	write_chunk(out, crc, "IHDR", ihdr);
	// raw image: per scan line one filter byte (0 = none) + RGB
	const size_t stride = static_cast<size_t>(width) * 3 + 1;
	std::vector<unsigned char> raw(stride * height);
	for (uint32_t y = 0; y < height; ++y) {
		unsigned char* row = &raw[stride * y];
		row[0] = 0;
		for (uint32_t x = 0; x < width; ++x) {
			const float* p = &rgba[4 * (static_cast<size_t>(y) * width + x)];
			for (int c = 0; c < 3; ++c) {
				float v = p[c];
				v = (v != v) ? 0.0f : std::min(std::max(v, 0.0f), 1.0f);
				row[1 + 3 * x + c] = static_cast<unsigned char>(v * 255.0f + 0.5f);
			}
		}
	}
	// zlib stream of stored blocks (at most 65535 bytes each) + Adler-32 of the raw bytes
	std::vector<unsigned char> z;
	z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
	z.push_back(0x78);
	z.push_back(0x01);
	uint32_t a = 1, b = 0;
	for (size_t off = 0; off < raw.size();) {
		const size_t n = std::min<size_t>(65535, raw.size() - off);
		z.push_back(off + n == raw.size() ? 1 : 0);
		z.push_back(static_cast<unsigned char>(n & 0xff));
		z.push_back(static_cast<unsigned char>(n >> 8));
		z.push_back(static_cast<unsigned char>(~n & 0xff));
		z.push_back(static_cast<unsigned char>((~n >> 8) & 0xff));
		z.insert(z.end(), raw.begin() + static_cast<std::ptrdiff_t>(off), raw.begin() + static_cast<std::ptrdiff_t>(off + n));
		for (size_t i = 0; i < n; ++i) {
			a = (a + raw[off + i]) % 65521u;
			b = (b + a) % 65521u;
		}
		off += n;
	}
	put_be32(z, (b << 16) | a);
	write_chunk(out, crc, "IDAT", z);
	write_chunk(out, crc, "IEND", {});
	return out.close();
	});
}

int tyr_write_pfm(const char* path, const float* rgba, uint32_t width, uint32_t height) {
	if (!path || !rgba || width == 0 || height == 0)
		return TYR_ERR_INVALID;
	return guarded([&]() -> int {
	OutFile out(path);
	if (!out.fp)
		return TYR_ERR_IO;
	char head[64];
	const int hn = std::snprintf(head, sizeof head, "PF\n%u %u\n-1.0\n", width, height); // negative scale = little endian
	out.write(head, 1, static_cast<size_t>(hn));
	std::vector<float> row(static_cast<size_t>(width) * 3);
	for (uint32_t y = height; y-- > 0;) { // PFM rows run bottom to top
		for (uint32_t x = 0; x < width; ++x)
			for (int c = 0; c < 3; ++c)
				row[3 * x + c] = rgba[4 * (static_cast<size_t>(y) * width + x) + c];
		out.write(row.data(), sizeof(float), row.size());
	}
	return out.close();
	});
}

} // extern "C"
