"""Mesh sink used inside Blender: the job of SF3D.import_mesh_blender
(/root/reference/StableFast/sf3d/system.py:528-598), with the per-loop Python UV loop (system.py:539-545) replaced by
one foreach_set and the image pixel upload done from contiguous arrays.  Imported only when `bpy` is importable."""
import numpy as np


def _image(bpy, name, pil_image, non_color=False):
    data = np.flip(np.array(pil_image), axis=0)
    img = bpy.data.images.new(name, width=pil_image.width, height=pil_image.height)
    img.pixels.foreach_set((data.astype(np.float32) / 255.0).ravel())
    if non_color:
        img.colorspace_settings.name = "Non-Color"
    return img


def import_mesh_blender(mesh, mesh_name="GeneratedMesh"):
    import bpy

    mesh_data = bpy.data.meshes.new(mesh_name)
    mesh_data.from_pydata(mesh["vertices"], [], mesh["faces"])
    obj = bpy.data.objects.new(name=mesh_name, object_data=mesh_data)
    bpy.context.collection.objects.link(obj)
    bpy.context.view_layer.objects.active = obj
    obj.select_set(True)
    if mesh.get("uvs") is not None:
        mesh_data.uv_layers.new(name="UVMap")
        loop_vert = np.empty(len(mesh_data.loops), np.int32)
        mesh_data.loops.foreach_get("vertex_index", loop_vert)
        mesh_data.uv_layers.active.data.foreach_set("uv", np.asarray(mesh["uvs"], np.float32)[loop_vert].ravel())
    material = bpy.data.materials.new(name="PBRMaterial")
    material.use_nodes = True
    obj.data.materials.append(material)
    nodes, links = material.node_tree.nodes, material.node_tree.links
    nodes.clear()
    bsdf = nodes.new(type="ShaderNodeBsdfPrincipled")
    bsdf.location = (0, 0)
    output = nodes.new(type="ShaderNodeOutputMaterial")
    links.new(bsdf.outputs["BSDF"], output.inputs["Surface"])
    if mesh.get("basecolor_tex"):
        tex = nodes.new("ShaderNodeTexImage")
        tex.image = _image(bpy, "BaseColor", mesh["basecolor_tex"])
        links.new(tex.outputs["Color"], bsdf.inputs["Base Color"])
    if mesh.get("roughness"):
        bsdf.inputs["Roughness"].default_value = mesh["roughness"]
    if mesh.get("metallic"):
        bsdf.inputs["Metallic"].default_value = mesh["metallic"]
    if mesh.get("bump_tex"):
        nm_tex = nodes.new("ShaderNodeTexImage")
        nm_tex.image = _image(bpy, "Bump", mesh["bump_tex"], non_color=True)
        nm = nodes.new("ShaderNodeNormalMap")
        links.new(nm_tex.outputs["Color"], nm.inputs["Color"])
        links.new(nm.outputs["Normal"], bsdf.inputs["Normal"])
    return obj
