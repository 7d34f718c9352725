"""Mesh sink used inside Blender: the job of TSR.import_obj_blender
(/root/reference/TripoSR/tsr/system.py:127-168), with the per-loop Python colour loop
(system.py:143-146) replaced by one foreach_set.  Imported only when `bpy` is importable."""
import numpy as np


def import_obj_blender(verts, faces, vertex_colors=None, name="NewMesh"):
    import bpy

    mesh_data = bpy.data.meshes.new(name=name)
    mesh_data.from_pydata(verts.tolist(), [], faces.tolist())
    new_object = bpy.data.objects.new(name=name, object_data=mesh_data)
    bpy.context.collection.objects.link(new_object)
    if vertex_colors is None:
        return new_object
    if vertex_colors.shape[1] == 3:
        vertex_colors = np.hstack((vertex_colors, np.ones((vertex_colors.shape[0], 1), vertex_colors.dtype)))
    layer_name = "%s_VC" % name
    mesh_data.vertex_colors.new(name=layer_name)
    color_layer = mesh_data.vertex_colors[layer_name]
    loop_vert = np.empty(len(mesh_data.loops), np.int32)
    mesh_data.loops.foreach_get("vertex_index", loop_vert)
    color_layer.data.foreach_set("color", vertex_colors[loop_vert].astype(np.float32).ravel())
    mat = bpy.data.materials.new(name="VertexColorMaterial")
    mesh_data.materials.append(mat)
    mat.use_nodes = True
    nodes, links = mat.node_tree.nodes, mat.node_tree.links
    for node in list(nodes):
        nodes.remove(node)
    out_node = nodes.new(type="ShaderNodeOutputMaterial")
    bsdf = nodes.new(type="ShaderNodeBsdfPrincipled")
    vc = nodes.new(type="ShaderNodeVertexColor")
    vc.layer_name = layer_name
    links.new(vc.outputs["Color"], bsdf.inputs["Base Color"])
    links.new(bsdf.outputs["BSDF"], out_node.inputs["Surface"])
    bsdf.inputs["Roughness"].default_value = 1
    bsdf.inputs["IOR"].default_value = 1.00
    return new_object
