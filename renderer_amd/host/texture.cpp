// texture.cpp — Texture (Texture.h:14-42, Texture.cpp:5-176) on the host: file loading, XML and the
// description handed to the device. The lookups themselves (Texture::getColor / getGrad) run on the
// device only: mirt_dev.h tex_color / tex_grad.
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "host.h"

namespace mirt_host {

// Texture::loadFile (Texture.cpp:5-41) accepts only names ending in "pfm" and leaves the texture untouched
// otherwise. Header as that loader consumes it: "P", one type character ('F' rgb, 'f' gray), two integers,
// one real (scale / endianness: read, ignored), then exactly one byte, or two when the first is a carriage
// return; the texels follow as raw fp32 in file order (no flip, no byte swap).
namespace {
struct PfmCursor {
    const std::vector<unsigned char>& buf;
    size_t at = 0;
    explicit PfmCursor(const std::vector<unsigned char>& b) : buf(b) {}
    void skipSpace() {
        while (at < buf.size() && isspace(buf[at])) at++;
    }
    bool number(double& out) {  // what fscanf's %d / %f would take after optional white space
        skipSpace();
        const char* begin = reinterpret_cast<const char*>(buf.data()) + at;
        char* end = nullptr;
        out = strtod(begin, &end);
        if (end == begin) return false;
        at += (size_t)(end - begin);
        return true;
    }
};
}  // namespace

void Texture::loadFile(const std::string& fileName) {
    const size_t L = fileName.size();
    if (L < 3 || fileName.compare(L - 3, 3, "pfm") != 0) return;
    std::vector<unsigned char> bytes;
    {
        FILE* f = fopen(fileName.c_str(), "rb");
        if (!f) throw std::runtime_error("Texture: cannot open " + fileName);
        unsigned char chunk[1 << 16];
        size_t got;
        while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) bytes.insert(bytes.end(), chunk, chunk + got);
        fclose(f);
    }
    bytes.push_back(0);  // terminator for strtod
    const std::runtime_error bad("Texture: " + fileName + " is not a readable .pfm");
    if (bytes.size() < 4 || bytes[0] != 'P' || (bytes[1] != 'F' && bytes[1] != 'f')) throw bad;
    const bool rgb = bytes[1] == 'F';
    PfmCursor cur(bytes);
    cur.at = 2;
    double w = 0, h = 0, scale = 0;
    if (!cur.number(w) || !cur.number(h) || !cur.number(scale)) throw bad;
    if (w < 1 || h < 1 || w != floor(w) || h != floor(h) || w * h > double(1 << 28)) throw bad;
    cur.at += (cur.at < bytes.size() && bytes[cur.at] == 0x0D) ? 2 : 1;
    const size_t count = (size_t)w * (size_t)h, floats = count * (rgb ? 3 : 1);
    if (cur.at + floats * sizeof(float) > bytes.size() - 1) throw bad;
    std::vector<float> raw(floats);
    memcpy(raw.data(), bytes.data() + cur.at, floats * sizeof(float));
    hasColor = rgb;
    width = (int)w;
    height = (int)h;
    colors.clear();
    values.clear();
    if (rgb) {
        colors.reserve(count);
        for (size_t k = 0; k < count; k++) colors.emplace_back(raw[3 * k], raw[3 * k + 1], raw[3 * k + 2]);
    } else {
        values.swap(raw);
    }
}

// Texture::toGrayScale (Texture.cpp:164-176): one channel per texel, the plain mean of r, g, b; a texture that
// already holds gray values keeps them (and drops any rgb copy).
void Texture::toGrayScale() {
    hasColor = false;
    if (values.empty() && !colors.empty()) {
        values.reserve(colors.size());
        for (const vec3f& c : colors) values.push_back((c.x + c.y + c.z) / 3);
    }
    colors.clear();
}

// Texture.cpp:149-162. <colorTransform> sets the COORDINATE transform there (Texture.cpp:154-155 call
// setCoordTransform for both tags); kept, so that a scene file means here what it means there.
void Texture::loadTextureFromXML(const ConfigManager* cm, XmlNode* nodeTex) {
    if (!nodeTex->first_node("path")) throw std::runtime_error("Texture: <path> missing");
    loadFile(cm->getFullPath(nodeTex->first_node("path")->value()));
    if (nodeTex->first_node("coordTransform")) setCoordTransform(readMatrix(nodeTex->first_node("coordTransform")->value()));
    if (nodeTex->first_node("colorTransform")) setCoordTransform(readMatrix(nodeTex->first_node("colorTransform")->value()));
    if (nodeTex->first_node("colorScale")) {
        const float scale = (float)atof(nodeTex->first_node("colorScale")->value());
        matrix4 sm;
        sm(0, 0) = sm(1, 1) = sm(2, 2) = scale;
        setColorTransform(colorTransform * sm);
    }
}

MirtTexture Texture::describe() const {
    MirtTexture d;
    memset(&d, 0, sizeof d);
    d.width = width;
    d.height = height;
    d.channels = hasColor ? 3 : 1;
    d.texels = hasColor ? &colors[0].x : values.data();
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            d.coord_transform[4 * r + c] = coordTransform(r, c);
            d.color_transform[4 * r + c] = colorTransform(r, c);
        }
    return d;
}

}  // namespace mirt_host
