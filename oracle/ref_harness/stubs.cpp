// TEST INFRASTRUCTURE ONLY. Empty bodies for the OpenGL-viewer entry points the
// reference defines in its viewport.cpp (which needs GLUT/GLU, absent here).
// None of them is on the rendering path.
#include "renderer.h"
#include "objects.h"
#include "lights.h"
#include "materials.h"
#include "texture.h"

void ShowViewport(Renderer *, bool) {}
void Sphere::ViewportDisplay(Material const *) const {}
void Plane::ViewportDisplay(Material const *) const {}
void TriObj::ViewportDisplay(Material const *) const {}
void GenLight::SetViewportParam(int, ColorA const &, ColorA const &, Vec4f const &) const {}
void PointLight::SetViewportLight(int) const {}
void PointLight::ViewportDisplay(Material const *) const {}
void MtlPhong::SetViewportMaterial(int) const {}
void MtlBlinn::SetViewportMaterial(int) const {}
void MtlMicrofacet::SetViewportMaterial(int) const {}
bool TextureFile::SetViewportTexture() const { return false; }
bool TextureChecker::SetViewportTexture() const { return false; }
