// Textures: the device form of TextureMap::Eval / TexturedColor::Eval / EvalEnvironment
// (src/scene.h:294-402) with the checkerboard and the bilinear, tiled image lookup
// (src/texture.cpp:100-134).  Same expressions in the same order as the reference.
#pragma once
#include "dmath.h"
#include "scene_dev.h"

namespace rtd {

RT_DEV V3 tile_clamp(V3 uvw) // scene.h:322-332
{
    V3 u = mk(uvw.x - (float)(int)uvw.x, uvw.y - (float)(int)uvw.y, uvw.z - (float)(int)uvw.z);
    if (u.x < 0) u.x += 1;
    if (u.y < 0) u.y += 1;
    if (u.z < 0) u.z += 1;
    return u;
}

RT_DEV V3 texel(const uint8_t *px, int width, int x, int y)
{
    const uint8_t *p = px + ((size_t)y * width + x) * 3;
    return mk((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f); // Color24::ToColor
}

RT_DEV V3 texture_eval(const DTexture &t, V3 uvw_in)
{
    const V3 u = tile_clamp(m34_mul(t.itm, uvw_in)); // TextureMap: TransformTo(uvw)
    if (t.type == RT_TEX_CHECKER) return ((u.x <= 0.5f) != (u.y <= 0.5f)) ? ld3(t.color2) : ld3(t.color1);
    const int width = t.width, height = t.height;
    if (width + height == 0) return mk(0, 0, 0);
    const float x = width * u.x;
    const float y = height * u.y;
    int ix = (int)x;
    int iy = (int)y;
    const float fx = x - ix;
    const float fy = y - iy;
    if (ix < 0) ix -= (ix / width - 1) * width;
    if (ix >= width) ix -= (ix / width) * width;
    int ixp = ix + 1;
    if (ixp >= width) ixp -= width;
    if (iy < 0) iy -= (iy / height - 1) * height;
    if (iy >= height) iy -= (iy / height) * height;
    int iyp = iy + 1;
    if (iyp >= height) iyp -= height;
    V3 c = texel(t.pixels, width, ix, iy) * ((1 - fx) * (1 - fy));
    c = c + texel(t.pixels, width, ixp, iy) * (fx * (1 - fy));
    c = c + texel(t.pixels, width, ix, iyp) * ((1 - fx) * fy);
    c = c + texel(t.pixels, width, ixp, iyp) * (fx * fy);
    return c;
}

// TexturedColor::Eval (scene.h:386): value * map, or the value alone.  tex = 1 + index, 0 = none
RT_DEV V3 textured_color(const DScene &S, const float *value, int tex, V3 uvw)
{
    if (tex <= 0) return ld3(value);
    return ld3(value) * texture_eval(S.textures[tex - 1], uvw);
}

// TexturedColor::EvalEnvironment (scene.h:390-397); |x| + |y| and the divisions run in double there
RT_DEV V3 textured_environment(const DScene &S, const float *value, int tex, V3 dir)
{
    if (tex <= 0) return ld3(value);
    const float l = len(dir);
    const float z = asinf(-dir.z / l) / (float)3.141592653589793238462643383279502884197169 + 0.5f;
    const double s = (double)fabsf(dir.x) + (double)fabsf(dir.y);
    const float x = (float)((double)dir.x / s);
    const float y = (float)((double)dir.y / s);
    return textured_color(S, value, tex, mk(0.5f + 0.5f * z * (x - y), 0.5f + 0.5f * z * (x + y), 0.0f));
}

} // namespace rtd
