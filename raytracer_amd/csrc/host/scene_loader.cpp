// scene.xml -> flattened scene.  Follows the reference's loader rule for rule
// (src/xmlload.cpp:41-270, src/xmlload.h:58-63) so that a scene file means the
// same thing on both sides of the C ABI: same defaults, same attribute
// handling, same order of transform composition, nodes in depth-first order.
#include "png_reader.h"
#include "scene_host.h"
#include "xform.h"
#include "xml_reader.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/stat.h>

void rt_scene_host::refresh_desc()
{
    mesh_descs.clear();
    for (auto &m : meshes) mesh_descs.push_back(m->desc());
    desc.nodes = nodes.data(); desc.n_nodes = (uint32_t)nodes.size();
    desc.meshes = mesh_descs.data(); desc.n_meshes = (uint32_t)mesh_descs.size();
    desc.materials = materials.data(); desc.n_materials = (uint32_t)materials.size();
    desc.lights = lights.data(); desc.n_lights = (uint32_t)lights.size();
    for (size_t i = 0; i < textures.size(); ++i) textures[i].pixels = texture_pixels[i].empty() ? nullptr : texture_pixels[i].data();
    desc.textures = textures.data(); desc.n_textures = (uint32_t)textures.size();
}

namespace rt {
namespace {

// case-insensitive string equality, as the reference compares tags and type names (xmlload.h:46-52)
bool ieq(const char *a, const char *b)
{
    if (a == b) return true;
    if (!a || !b) return false;
    for (;; ++a, ++b) {
        if (tolower((unsigned char)*a) != tolower((unsigned char)*b)) return false;
        if (*a == '\0') return true;
    }
}

bool read_float(const XmlElement *e, float &f, const char *name = "value")
{
    if (!e) return false;
    const char *s = e->attr(name);
    if (!s) return false;
    float v;
    if (sscanf(s, "%f", &v) != 1) return false;
    f = v;
    return true;
}
bool read_int(const XmlElement *e, int &i, const char *name = "value")
{
    if (!e) return false;
    const char *s = e->attr(name);
    if (!s) return false;
    char *end;
    long v = strtol(s, &end, 0);
    if (end == s) return false;
    i = (int)v;
    return true;
}
// x,y,z attributes over a default, then multiplied by `value` when present
void read_vec3(const XmlElement *e, Vec3 &v, Vec3 def = {0, 0, 0})
{
    if (!e) return;
    v = def;
    read_float(e, v.x, "x"); read_float(e, v.y, "y"); read_float(e, v.z, "z");
    float f = 1;
    if (read_float(e, f)) v = v * f;
}
void read_color(const XmlElement *e, float c[3], Vec3 def = {1, 1, 1})
{
    if (!e) return;
    Vec3 v = def;
    read_float(e, v.x, "r"); read_float(e, v.y, "g"); read_float(e, v.z, "b");
    float f = 1;
    if (read_float(e, f)) v = v * f;
    c[0] = v.x; c[1] = v.y; c[2] = v.z;
}

struct Loader {
    rt_scene_host &S;
    std::string base_dir;
    std::string err;
    std::vector<std::string> node_material_names; // per node, "" = none
    std::vector<int> node_sub_materials;          // per node: 1 = an OBJ node without a material attribute whose file names materials

    bool file_exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }

    // channels both renderers read with GetValue() only: a texture there has no effect in the reference either
    void textured(const XmlElement *e, const char *what)
    {
        if (e && e->attr("texture"))
            S.warnings.push_back(std::string("texture on <") + what + "> has no effect: the renderers read this channel's value only");
    }

    // PPM (P6, maxval 255) as LoadPPM reads it (src/texture.cpp:36-58): magic line, comment lines, "W H", comment lines,
    // the maxval line, then W*H RGB bytes
    bool load_ppm(const std::string &path, int &w, int &h, std::vector<uint8_t> &px)
    {
        FILE *fp = fopen(path.c_str(), "rb");
        if (!fp) return false;
        auto line = [&](char *buf, int size) {
            int i = 0;
            for (; i < size; ++i) {
                const int c = fgetc(fp);
                if (c == EOF || c == '\n' || c == '\r') { buf[i] = '\0'; return; }
                buf[i] = (char)c;
            }
            buf[size - 1] = '\0';
        };
        char buf[1024];
        line(buf, sizeof buf);
        bool ok = buf[0] == 'P' && buf[1] == '6';
        if (ok) {
            line(buf, sizeof buf);
            while (buf[0] == '#') line(buf, sizeof buf);
            ok = sscanf(buf, "%d %d", &w, &h) == 2 && w > 0 && h > 0 && (int64_t)w * h <= (1 << 28);
        }
        if (ok) {
            line(buf, sizeof buf);
            while (buf[0] == '#') line(buf, sizeof buf);
            px.resize((size_t)w * h * 3);
            ok = fread(px.data(), 3, (size_t)w * h, fp) == (size_t)w * h;
        }
        fclose(fp);
        return ok;
    }

    // ReadTextureFile + TextureFile::LoadFile (xmlload.cpp:356-371, texture.cpp:62-96): the last three letters of the
    // name, whatever their case, say PNG or PPM; anything else, or a file that cannot be decoded, is no texture
    bool read_texture_file(const char *name, rt_texture &t, std::vector<uint8_t> &px)
    {
        const size_t len = strlen(name);
        std::string path = name;
        if (!file_exists(path) && !base_dir.empty()) path = base_dir + "/" + name;
        bool ok = false;
        std::string why;
        if (len >= 3 && ieq(name + len - 3, "ppm")) ok = load_ppm(path, t.width, t.height, px);
        else if (len >= 3 && ieq(name + len - 3, "png")) ok = load_png_rgb8(path, t.width, t.height, px, why);
        else why = "neither a PNG nor a PPM name";
        if (!ok) {
            S.warnings.push_back(std::string("cannot read texture file ") + name + (why.empty() ? "" : ": " + why));
            return false;
        }
        t.type = RT_TEX_IMAGE;
        return true;
    }

    // Loader::ReadTextureMap (xmlload.cpp:326-344): "checkerboard" with <color1>/<color2>, or an image file; the element's
    // scale / rotate / translate children are the texture transformation.  Returns 1 + texture index, 0 = no texture.
    int read_texture_map(const XmlElement *e)
    {
        const char *name = e ? e->attr("texture") : nullptr;
        if (!name) return 0;
        rt_texture t{};
        std::vector<uint8_t> px;
        if (ieq(name, "checkerboard")) {
            t.type = RT_TEX_CHECKER;
            t.color1[0] = t.color1[1] = t.color1[2] = 0.0f; // texture.h:50
            t.color2[0] = t.color2[1] = t.color2[2] = 1.0f;
            read_color(e->child("color1"), t.color1);
            read_color(e->child("color2"), t.color2);
            if ((e->child("color1") && e->child("color1")->attr("texture")) || (e->child("color2") && e->child("color2")->attr("texture")))
                S.warnings.push_back("textures inside a checkerboard's colours are not supported: their values are used");
        } else if (!read_texture_file(name, t, px)) return 0; // the reference prints an error and goes on without the texture (xmlload.cpp:337-339,356-371)
        Mat34 tm = Mat34::identity(), itm = Mat34::identity();
        load_transform(*e, tm, itm);
        memcpy(t.itm, itm.c, sizeof t.itm);
        S.textures.push_back(t);
        S.texture_pixels.push_back(std::move(px));
        return (int)S.textures.size();
    }

    // Transformation::Load (xmlload.cpp:173-193): each child element pre-multiplies the node matrix
    void load_transform(const XmlElement &e, Mat34 &tm, Mat34 &itm)
    {
        for (auto &c : e.children) {
            Mat34 m;
            if (ieq(c->tag.c_str(), "scale")) {
                Vec3 s; read_vec3(c.get(), s, {1, 1, 1});
                m = Mat34::scale(s);
            } else if (ieq(c->tag.c_str(), "rotate")) {
                Vec3 s{0, 0, 0}; read_vec3(c.get(), s);
                s = normalized(s);
                float a = 0.0f; read_float(c.get(), a, "angle");
                const float pi = (float)3.141592653589793238462643383279502884197169;
                m = Mat34::rotation(s, a * (pi / 180.0f));
            } else if (ieq(c->tag.c_str(), "translate")) {
                Vec3 t{0, 0, 0}; read_vec3(c.get(), t);
                m = Mat34::translation(t);
            } else continue;
            tm = m * tm;
            itm = tm.inverse();
        }
    }

    int find_mesh(const std::string &name)
    {
        for (size_t i = 0; i < S.meshes.size(); ++i) if (S.meshes[i]->name == name) return (int)i;
        return -1;
    }

    // LoadNode (xmlload.cpp:124-169)
    bool load_node(const XmlElement &e, int parent)
    {
        rt_node n{};
        Mat34 tm = Mat34::identity(), itm = Mat34::identity();
        n.parent = parent; n.prim = RT_PRIM_NONE; n.mesh = -1; n.material = -1;
        const char *name = e.attr("name");
        const char *mtl = e.attr("material");
        const char *type = e.attr("type");
        if (type) {
            if (ieq(type, "sphere")) n.prim = RT_PRIM_SPHERE;
            else if (ieq(type, "plane")) n.prim = RT_PRIM_PLANE;
            else if (ieq(type, "obj")) {
                const std::string nm = name ? name : "";
                int mi = find_mesh(nm);
                if (mi < 0) {
                    auto m = std::make_unique<HostMesh>();
                    m->name = nm;
                    std::string path = nm;
                    if (!file_exists(path) && !base_dir.empty()) path = base_dir + "/" + nm;
                    if (file_exists(path)) {
                        if (!load_obj(path, *m, err)) return false;
                        build_bvh(m->v.data(), m->f.data(), m->nf(), 4, m->bvh, m->elements, m->depth);
                    }
                    // a mesh whose file is absent stays empty until rt_scene_host_set_mesh fills it
                    S.meshes.push_back(std::move(m));
                    mi = (int)S.meshes.size() - 1;
                }
                n.prim = RT_PRIM_MESH;
                n.mesh = mi;
            } else S.warnings.push_back(std::string("unknown object type ") + type);
        }
        load_transform(e, tm, itm);
        memcpy(n.tm, tm.c, sizeof n.tm);
        memcpy(n.itm, itm.c, sizeof n.itm);
        const int me = (int)S.nodes.size();
        S.nodes.push_back(n);
        S.node_names.push_back(name ? name : "");
        node_material_names.push_back(mtl ? mtl : "");
        // xmlload.cpp:157: such a node gets the materials of its OBJ file (decided when the file is loaded: a mesh filled in
        // later through rt_scene_host_set_mesh has none)
        node_sub_materials.push_back(!mtl && n.prim == RT_PRIM_MESH && !S.meshes[(size_t)n.mesh]->mtls.empty() ? 1 : 0);
        for (auto &c : e.children)
            if (ieq(c->tag.c_str(), "object") && !load_node(*c, me)) return false;
        return true;
    }

    void load_light(const XmlElement &e)
    {
        const char *type = e.attr("type");
        rt_light l{};
        if (ieq(type, "ambient")) {
            l.type = RT_LIGHT_AMBIENT;
            read_color(e.child("intensity"), l.intensity);
        } else if (ieq(type, "direct")) {
            l.type = RT_LIGHT_DIRECT;
            read_color(e.child("intensity"), l.intensity);
            Vec3 d{0, 0, 0}; read_vec3(e.child("direction"), d);
            d = normalized(d);
            l.direction[0] = d.x; l.direction[1] = d.y; l.direction[2] = d.z;
        } else if (ieq(type, "point")) {
            l.type = RT_LIGHT_POINT;
            read_color(e.child("intensity"), l.intensity);
            Vec3 p{0, 0, 0}; read_vec3(e.child("position"), p);
            l.position[0] = p.x; l.position[1] = p.y; l.position[2] = p.z;
            read_float(e.child("size"), l.size);
            read_float(e.child("attenuation"), l.attenuation);
        } else {
            S.warnings.push_back(std::string("unknown light type ") + (type ? type : "(none)"));
            return;
        }
        S.lights.push_back(l);
    }

    // LoadMaterial + MtlBasePhongBlinn::Load (xmlload.cpp:242-270), defaults from materials.h:53-60
    void load_material(const XmlElement &e)
    {
        const char *type = e.attr("type");
        if (!ieq(type, "blinn")) {
            S.warnings.push_back(std::string("unknown material type ") + (type ? type : "(none)"));
            return;
        }
        rt_material m{};
        for (int i = 0; i < 3; ++i) { m.diffuse[i] = 0.5f; m.specular[i] = 0.7f; }
        m.glossiness = 20.0f;
        m.ior = 1.5f;
        read_color(e.child("diffuse"), m.diffuse);       m.diffuse_texture = read_texture_map(e.child("diffuse"));
        read_color(e.child("specular"), m.specular);     textured(e.child("specular"), "specular");
        if (const XmlElement *g = e.child("glossiness")) { float f = 1.0f; read_float(g, f); m.glossiness = f; textured(g, "glossiness"); }
        read_color(e.child("emission"), m.emission);     textured(e.child("emission"), "emission");
        read_color(e.child("reflection"), m.reflection); textured(e.child("reflection"), "reflection");
        read_color(e.child("refraction"), m.refraction); textured(e.child("refraction"), "refraction");
        read_float(e.child("refraction"), m.ior, "index");
        read_color(e.child("absorption"), m.absorption);
        S.materials.push_back(m);
        const char *name = e.attr("name");
        S.material_names.push_back(name ? name : "");
    }

    // SetNodeMaterials + CreateMultiMtl (xmlload.cpp:278-286,298-322): the materials of an OBJ file become one MultiMtl
    // named after the node, looked up by that name first (a material of the scene file, or an earlier node's, with the
    // same name wins).  Here the MultiMtl is its members as consecutive materials; the first carries the name.
    // Returns the index of the first member.
    int multi_material(const std::string &node_name, HostMesh &mesh)
    {
        for (size_t k = 0; k < S.material_names.size(); ++k)
            if (S.material_names[k] == node_name) {
                if (!mesh.members_created) { // an ordinary material of that name: it has no members for the faces to pick from
                    S.warnings.push_back("material " + node_name + " stands in for the materials of the OBJ file of the same name");
                    mesh.face_mtl.clear();
                }
                return (int)k;
            }
        mesh.members_created = true;
        const int first = (int)S.materials.size();
        for (size_t i = 0; i < mesh.mtls.size(); ++i) {
            const ObjMaterial &o = mesh.mtls[i];
            rt_material m{};
            for (int c = 0; c < 3; ++c) { m.diffuse[c] = o.Kd[c]; m.specular[c] = o.Ks[c]; }
            m.glossiness = o.Ns;
            m.ior = o.Ni;
            // (the second line of the reference sets the DIFFUSE texture again, from map_Ks: xmlload.cpp:310)
            const std::string &map = !o.map_Ks.empty() ? o.map_Ks : o.map_Kd;
            if (!map.empty()) m.diffuse_texture = obj_texture(map);
            if (o.illum > 2 && o.illum <= 7) {
                for (int c = 0; c < 3; ++c) m.reflection[c] = o.Ks[c];
                if (o.illum >= 6) for (int c = 0; c < 3; ++c) m.refraction[c] = 1 - o.Tf[c];
            }
            S.materials.push_back(m);
            S.material_names.push_back(i == 0 ? node_name : node_name + "#" + std::to_string(i));
        }
        return first;
    }
    // new TextureMap(ReadTextureFile(...)) (xmlload.cpp:309-310): identity transformation; a file that cannot be read
    // leaves a map without a texture, which evaluates to black (scene.h:350) - here an image of no pixels
    int obj_texture(const std::string &name)
    {
        rt_texture t{};
        std::vector<uint8_t> px;
        if (!read_texture_file(name.c_str(), t, px)) { t.type = RT_TEX_IMAGE; t.width = t.height = 0; px.clear(); }
        memcpy(t.itm, Mat34::identity().c, sizeof t.itm);
        S.textures.push_back(t);
        S.texture_pixels.push_back(std::move(px));
        return (int)S.textures.size();
    }

    // Camera::Load (xmlload.cpp:104-120), defaults from scene.h:466-477
    void load_camera(const XmlElement &e)
    {
        Vec3 pos{0, 0, 0}, dir{0, 0, -1}, up{0, 1, 0};
        rt_camera &c = S.camera;
        c.fov = 40; c.focaldist = 1; c.dof = 0; c.width = 1920; c.height = 1080;
        read_vec3(e.child("position"), pos);
        read_vec3(e.child("target"), dir);
        read_vec3(e.child("up"), up);
        read_float(e.child("fov"), c.fov);
        read_float(e.child("focaldist"), c.focaldist);
        read_float(e.child("dof"), c.dof);
        read_int(e.child("width"), c.width);
        read_int(e.child("height"), c.height);
        dir = dir - pos;
        dir = normalized(dir);
        const Vec3 x = cross(dir, up);
        up = normalized(cross(x, dir));
        c.pos[0] = pos.x; c.pos[1] = pos.y; c.pos[2] = pos.z;
        c.dir[0] = dir.x; c.dir[1] = dir.y; c.dir[2] = dir.z;
        c.up[0] = up.x; c.up[1] = up.y; c.up[2] = up.z;
        c.srgb = ieq(e.attr("gamma"), "sRGB") ? 1 : 0;
    }

    bool load(const std::string &xml)
    {
        std::unique_ptr<XmlElement> doc = xml_parse(xml, err);
        if (!doc) return false;
        const XmlElement *x = doc->child("xml");
        if (!x) { err = "no \"xml\" tag found"; return false; }
        const XmlElement *xscene = x->child("scene");
        if (!xscene) { err = "no \"scene\" tag found"; return false; }
        const XmlElement *xcam = x->child("camera");
        if (!xcam) { err = "no \"camera\" tag found"; return false; }

        // root node: identity transform, no object (scene.h:424,497)
        rt_node root{};
        memcpy(root.tm, Mat34::identity().c, sizeof root.tm);
        memcpy(root.itm, Mat34::identity().c, sizeof root.itm);
        root.parent = -1; root.prim = RT_PRIM_NONE; root.mesh = -1; root.material = -1;
        S.nodes.push_back(root);
        S.node_names.push_back("");
        node_material_names.push_back("");
        node_sub_materials.push_back(0);

        // Scene::Load (xmlload.cpp:80-100). The reference appends every top-level
        // object to the root in file order; nested objects follow their parent, so a
        // depth-first listing needs the top-level objects visited in file order.
        for (auto &c : xscene->children) {
            const char *t = c->tag.c_str();
            if (ieq(t, "object")) { if (!load_node(*c, 0)) return false; }
            else if (ieq(t, "light")) load_light(*c);
            else if (ieq(t, "material")) load_material(*c);
            else if (ieq(t, "background")) { read_color(c.get(), S.desc.background); S.desc.background_texture = read_texture_map(c.get()); }
            else if (ieq(t, "environment")) { read_color(c.get(), S.desc.environment); S.desc.environment_texture = read_texture_map(c.get()); }
            else S.warnings.push_back(std::string("unknown tag ") + t);
        }
        // SetNodeMaterials (xmlload.cpp:274-294): names resolved after all materials are known, exact match
        for (size_t i = 0; i < S.nodes.size(); ++i) {
            if (node_sub_materials[i]) { S.nodes[i].material = multi_material(S.node_names[i], *S.meshes[(size_t)S.nodes[i].mesh]); continue; }
            const std::string &mn = node_material_names[i];
            if (mn.empty()) continue;
            for (size_t k = 0; k < S.material_names.size(); ++k)
                if (S.material_names[k] == mn) { S.nodes[i].material = (int)k; break; }
        }
        load_camera(*xcam);
        S.refresh_desc();
        return true;
    }
};

} // namespace

bool load_scene_xml(const std::string &xml, const std::string &base_dir, rt_scene_host &out, std::string &err)
{
    Loader L{out, base_dir, {}, {}, {}};
    // TexturedColor default-constructs to black (scene.h:376)
    for (int i = 0; i < 3; ++i) out.desc.background[i] = out.desc.environment[i] = 0.0f;
    if (!L.load(xml)) { err = L.err; return false; }
    return true;
}

} // namespace rt
