// texture.cpp — Texture (Texture.h:14-42, Texture.cpp:5-176) on the host: file loading, XML and the
// description handed to the device. The lookups that matter run in mirt_dev.h (tex_color); getColor
// here is the same arithmetic for host-side checks.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "host.h"

namespace mirt_host {

// Texture::loadFile (Texture.cpp:5-41): ".pfm" only, "PF" = rgb / "Pf" = gray, texels kept in file order,
// the scale / endianness field read and ignored, one byte (two after a CR) between header and data.
// Any other extension leaves the texture as it was, like the reference.
void Texture::loadFile(const std::string& fileName) {
    if (fileName.size() < 3 || fileName.substr(fileName.size() - 3, 3) != "pfm") return;
    FILE* file = fopen(fileName.c_str(), "rb");
    if (!file) throw std::runtime_error("Texture: cannot open " + fileName);
    char c = 0;
    int w = 0, h = 0;
    float form = 0.f;
    bool ok = fscanf(file, "P%c", &c) == 1 && (c == 'f' || c == 'F') && fscanf(file, "%d %d", &w, &h) == 2 &&
              w > 0 && h > 0 && (long long)w * h <= (1ll << 28) && fscanf(file, "%f", &form) == 1;
    if (ok) {
        hasColor = c == 'F';
        ok = fread(&c, 1, 1, file) == 1;
        if (ok && c == 0x0D) ok = fread(&c, 1, 1, file) == 1;
    }
    if (ok) {
        width = w;
        height = h;
        const size_t n = (size_t)w * h;
        if (hasColor) {
            std::vector<float> raw(3 * n);
            ok = fread(raw.data(), sizeof(float), raw.size(), file) == raw.size();
            colors.resize(n);
            values.clear();
            for (size_t i = 0; i < n; i++) colors[i] = vec3f(raw[3 * i], raw[3 * i + 1], raw[3 * i + 2]);
        } else {
            values.resize(n);
            colors.clear();
            ok = fread(values.data(), sizeof(float), n, file) == n;
        }
    }
    fclose(file);
    if (!ok) throw std::runtime_error("Texture: " + fileName + " is not a readable .pfm");
}

// Texture.cpp:164-176
void Texture::toGrayScale() {
    hasColor = false;
    if (colors.empty() || !values.empty()) {
        colors.clear();
        return;
    }
    values.resize(colors.size());
    for (size_t i = 0; i < colors.size(); i++) values[i] = (colors[i].x + colors[i].y + colors[i].z) / 3;
    colors.clear();
}

// Texture.cpp:43-94 (the gray branch reads its fourth texel at (top, left), Texture.cpp:86)
vec3f Texture::getColor(const vec3f& coord) const {
    vec3f tc = coordTransform.apply(coord, 1.f);
    float u = tc.x * width - 0.5;
    float v = tc.y * height - 0.5;
    int left = (int)floor(u);
    int right = left + 1;
    int bottom = (int)floor(v);
    int top = bottom + 1;
    u -= left;
    v -= bottom;
    left %= width;
    right %= width;
    top %= height;
    bottom %= height;
    if (left < 0) left += width;
    if (right < 0) right += width;
    if (top < 0) top += height;
    if (bottom < 0) bottom += height;
    if (hasColor) {
        vec3f color(0, 0, 0);
        color = color + colors[bottom * width + left] * (1 - u) * (1 - v);
        color = color + colors[bottom * width + right] * u * (1 - v);
        color = color + colors[top * width + left] * (1 - u) * v;
        color = color + colors[top * width + right] * u * v;
        return colorTransform.apply(color, 1.f);
    }
    float color = 0;
    color += values[bottom * width + left] * (1 - u) * (1 - v);
    color += values[bottom * width + right] * u * (1 - v);
    color += values[top * width + left] * (1 - u) * v;
    color += values[top * width + left] * u * v;
    return colorTransform.apply(vec3f(color, color, color), 1.f);
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
