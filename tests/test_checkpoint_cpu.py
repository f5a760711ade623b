"""The two-stage recipe of the fusion detector at state-dict level: a camera-only run (cam_stream/LSS.py) and a
radar-only run (radar_stream/pointpillars_4DRadar.py) stitched into the fusion model (bevfusion.py:288-290) by the
rules of the reference's tools/train.py:270-425."""
import copy

import torch


def _randomise(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for v in model.state_dict().values():
            if v.is_floating_point():
                v.copy_(torch.randn(v.shape, generator=g))
    return model


def test_stage1_checkpoints_stitch_into_the_fusion_detector(tmp_path):
    from omnihd_amd import harness
    from omnihd_amd.checkpoint import stitch_checkpoints
    from omnihd_amd.mm.config import build_detector
    base = harness.tiny_model_cfg(7)
    cam_cfg = copy.deepcopy(base)
    for k in ("pts_voxel_layer", "pts_voxel_encoder", "pts_middle_encoder", "pts_backbone", "pts_neck", "se"):
        cam_cfg.pop(k)
    cam_cfg.update(lc_fusion=False)
    cam_cfg["pts_bbox_head"].update(in_channels=256, feat_channels=256)
    cam = _randomise(build_detector(cam_cfg), 1)
    radar = _randomise(build_detector(harness.pillars_model_cfg(base, "radar")), 2)
    fusion = _randomise(build_detector(base), 3)
    before = {k: v.clone() for k, v in fusion.state_dict().items()}
    torch.save(dict(state_dict=cam.state_dict(), meta={}), tmp_path / "lss.pth")
    torch.save(dict(state_dict={"module." + k: v for k, v in radar.state_dict().items()}), tmp_path / "radar.pth")
    rep = stitch_checkpoints(fusion, dict(load_lift_from=str(tmp_path / "lss.pth"), load_from=str(tmp_path / "radar.pth"),
                                          resume_from=None))
    assert list(rep) == ["load_lift_from", "load_from"]
    assert not rep["load_lift_from"]["skipped"] and not rep["load_from"]["skipped"]
    sd, csd, rsd = fusion.state_dict(), cam.state_dict(), radar.state_dict()
    for k, v in sd.items():
        top = k.split(".")[0]
        if top in ("img_backbone", "img_neck", "lift_splat_shot_vis"):
            assert torch.equal(v, csd[k]), k                         # from the camera run
        elif top in ("pts_voxel_encoder", "pts_backbone", "pts_neck", "pts_bbox_head"):
            assert torch.equal(v, rsd[k]), k                         # from the radar run, head included
        else:
            assert top in ("reduc_conv", "seblock") and torch.equal(v, before[k]), k      # new in the fusion model
    # the camera run's head (256 channels) never reaches the fusion head (384 channels)
    assert csd["pts_bbox_head.conv_cls.weight"].shape != sd["pts_bbox_head.conv_cls.weight"].shape


def test_image_pretrain_rules_and_shape_mismatch_reporting(tmp_path):
    from omnihd_amd import harness
    from omnihd_amd.checkpoint import state_dict_of, stitch_checkpoints
    from omnihd_amd.mm.config import build_detector
    base = harness.tiny_model_cfg(7)
    fusion = _randomise(build_detector(base), 4)
    donor = _randomise(build_detector(base), 5).state_dict()
    # an image-detector checkpoint: backbone.* / neck.* / bbox_head.* names, wrapped as {'model': ...} with module. prefix
    img = {"module." + k.replace("img_backbone.", "backbone.").replace("img_neck.", "neck."): v for k, v in donor.items()
           if k.startswith(("img_backbone.", "img_neck."))}
    img["module.bbox_head.cls.weight"] = torch.zeros(3)
    torch.save(dict(model=img), tmp_path / "img.pth")
    # a point-cloud detector checkpoint: backbone / neck / voxel_encoder / bbox_head / middle_encoder names
    pts = {k.replace("pts_backbone.", "backbone.").replace("pts_neck.", "neck.").replace("pts_voxel_encoder.", "voxel_encoder."): v
           for k, v in donor.items() if k.startswith(("pts_backbone.", "pts_neck.", "pts_voxel_encoder."))}
    pts["bbox_head.conv_cls.weight"] = torch.zeros(2)
    pts["middle_encoder.dummy"] = torch.zeros(1)
    bad = next(k for k in pts if k.startswith("backbone.") and k.endswith("weight"))
    pts[bad] = torch.zeros(7)                                          # wrong shape: skipped, reported
    torch.save(pts, tmp_path / "pts.pth")
    assert state_dict_of(dict(model=img)) is img and state_dict_of(pts) is pts
    before = {k: v.clone() for k, v in fusion.state_dict().items()}
    rep = stitch_checkpoints(fusion, dict(load_img_from=str(tmp_path / "img.pth"), load_pts_from=str(tmp_path / "pts.pth")))
    sd = fusion.state_dict()
    assert rep["load_pts_from"]["skipped"] == [bad.replace("backbone.", "pts_backbone.")]
    assert not rep["load_img_from"]["skipped"] and all(k.startswith(("img_backbone.", "img_neck.")) for k in rep["load_img_from"]["loaded"])
    for k, v in sd.items():
        top = k.split(".")[0]
        if k == bad.replace("backbone.", "pts_backbone."):
            assert torch.equal(v, before[k])
        elif top in ("img_backbone", "img_neck", "pts_backbone", "pts_neck", "pts_voxel_encoder"):
            assert torch.equal(v, donor[k]), k
        else:
            assert torch.equal(v, before[k]), k
    # load_img_from_and_not_change_state_dict: same names, bbox_head dropped
    same = dict(state_dict={**{k: v for k, v in donor.items() if k.startswith("lift_splat_shot_vis.")}, "bbox_head.x": torch.zeros(1)})
    torch.save(same, tmp_path / "same.pth")
    rep = stitch_checkpoints(fusion, dict(load_img_from_and_not_change_state_dict=str(tmp_path / "same.pth")))
    assert "bbox_head.x" not in rep["load_img_from_and_not_change_state_dict"]["skipped"]
    assert all(torch.equal(fusion.state_dict()[k], donor[k]) for k in same["state_dict"] if k.startswith("lift"))
