"""The drop-in layout survives the reference scripts' OTHER imports.

The scripts import, next to ``models.vae_gan``, modules the engine does not provide (``configs.gan_config``,
``configs.data_config``, ``train.train_utils.evaluate``, ``data_preprocessing.data_loader``).  With
``thesis-fmri-reconstruction_amd`` first and the project root second on ``sys.path`` those must still resolve --
to the project's own files -- while ``models.vae_gan``, ``configs.models_config`` and the two metric classes
resolve to the engine.  The project here is a stub tree with the reference's layout (own dummy files, written by
this test); the import lines executed are the literal ones of
  train/train_vgan_stage1.py:21-25, train/train_wae_stage3.py:21-26, inference/inference_gan.py:19-26.
Runs in a child interpreter so that the overlay packages are imported fresh with the two-entry path.
"""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PKG = os.path.join(ROOT, "thesis-fmri-reconstruction_amd")

STUB = {
    "configs/__init__.py": "",
    "configs/gan_config.py": "batch_size = 64\nMARK = 'project gan_config'\n",
    "configs/data_config.py": "MARK = 'project data_config'\n",
    "configs/wae_config.py": "MARK = 'project wae_config'\n",
    "configs/inference_config.py": "batch_size = 16\nMARK = 'project inference_config'\n",
    # the project's own copies of the modules the engine replaces: must be SHADOWED
    "configs/models_config.py": "MARK = 'project models_config (must be shadowed)'\nimage_size = -1\n",
    "models/vae_gan.py": "raise ImportError('project models.vae_gan was imported: the engine must shadow it')\n",
    "models/other_net.py": "MARK = 'project models.other_net'\n",
    "train/__init__.py": "",
    "train/train_utils.py": textwrap.dedent("""
        MARK = 'project train_utils'
        def evaluate(model, dataloader, **kw):
            return 'project evaluate'
        def objective_assessment(model, dataloader, **kw):
            return 'project objective_assessment'
        class EarlyStopping(object):
            pass
        class PearsonCorrelation(object):       # must be shadowed by the engine's
            MARK = 'project PCC'
        class StructuralSimilarity(object):
            MARK = 'project SSIM'
        """),
    "train/other_script.py": "MARK = 'project train.other_script'\n",
    "data_preprocessing/__init__.py": "",
    "data_preprocessing/data_loader.py": textwrap.dedent("""
        import configs.data_config as data_cfg          # reference data_loader.py:13
        class _T(object):
            pass
        CocoDataloader = GreyToColor = BoldRoiDataloader = CenterCrop = Rescale = RandomShift = _T
        SampleToTensor = Normalization = _T
        def split_subject_data(*a, **k):
            return None
        """),
}

CHILD = textwrap.dedent("""
    import sys
    assert sys.path[1].endswith('thesis-fmri-reconstruction_amd'), sys.path[:3]

    # --- train/train_vgan_stage1.py:21-25 ---
    import configs.gan_config as gan_cfg
    import configs.data_config as data_cfg
    from models.vae_gan import VaeGan, WaeGan
    from data_preprocessing.data_loader import CocoDataloader, GreyToColor
    from train.train_utils import evaluate, PearsonCorrelation, StructuralSimilarity

    # --- train/train_wae_stage3.py:21-26 ---
    import configs.data_config as data_cfg
    import configs.wae_config as wae_cfg
    from models.vae_gan import WaeGan, CognitiveEncoder, WaeGanCognitive, Decoder
    from train.train_utils import evaluate, PearsonCorrelation, StructuralSimilarity
    from data_preprocessing.data_loader import BoldRoiDataloader, CenterCrop, Rescale, RandomShift, SampleToTensor, \\
        Normalization, split_subject_data

    # --- inference/inference_gan.py:19-26 ---
    import configs.inference_config as inf_cfg
    import configs.data_config as data_cfg
    from models.vae_gan import CognitiveEncoder, Encoder, Decoder, VaeGanCognitive, VaeGan, Discriminator, WaeGan, \\
        WaeGanCognitive
    from train.train_utils import evaluate, objective_assessment, PearsonCorrelation, StructuralSimilarity
    from data_preprocessing.data_loader import BoldRoiDataloader, Rescale, CenterCrop, SampleToTensor, RandomShift, \\
        Normalization, split_subject_data, CocoDataloader, GreyToColor

    # --- models/vae_gan.py:8 (what the engine's module itself reads) ---
    import configs.models_config as config

    import os
    import models.vae_gan, train.train_utils, configs
    pkg = sys.path[1]
    assert os.path.dirname(models.vae_gan.__file__) == os.path.join(pkg, 'models')
    assert os.path.dirname(config.__file__) == os.path.join(pkg, 'configs') and config.image_size == 100
    assert os.path.dirname(train.train_utils.__file__) == os.path.join(pkg, 'train')
    assert gan_cfg.MARK == 'project gan_config' and data_cfg.MARK == 'project data_config'
    assert wae_cfg.MARK == 'project wae_config' and inf_cfg.MARK == 'project inference_config'
    assert evaluate(None, None) == 'project evaluate'
    assert objective_assessment(None, None) == 'project objective_assessment'
    # the metric classes are the engine's (torch modules over csrc/metrics.hip), not the project's
    import torch
    assert issubclass(PearsonCorrelation, torch.nn.Module) and not hasattr(PearsonCorrelation, 'MARK')
    assert issubclass(StructuralSimilarity, torch.nn.Module) and not hasattr(StructuralSimilarity, 'MARK')
    assert train.train_utils.EarlyStopping.__module__ == 'train._shadowed_train_utils'
    # sibling modules of the overlaid packages still resolve to the project
    import train.other_script, models.other_net
    assert train.other_script.MARK == 'project train.other_script'
    assert models.other_net.MARK == 'project models.other_net'
    try:
        from train.train_utils import no_such_name
    except ImportError as e:
        assert 'no_such_name' in str(e)
    else:
        raise AssertionError('missing name did not raise')
    # the models build on the overlay exactly as the scripts build them (train_vgan_stage1.py:236)
    config.use_px64()
    m = VaeGan(device='cpu', z_size=128)
    assert len(m.state_dict()) > 0
    print('DROPIN_OK')
    """)


def _write_stub(root):
    for rel, text in STUB.items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text)


def _run_child(code, path_entries, cwd):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join(path_entries)
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    return subprocess.run([sys.executable, "-c", code], cwd=cwd, env=env, capture_output=True, text=True, timeout=600)


def test_reference_script_import_lines_resolve_on_the_overlay(tmp_path):
    proj = tmp_path / "project"
    _write_stub(str(proj))
    scratch = tmp_path / "cwd"
    scratch.mkdir()
    r = _run_child(CHILD, [PKG, str(proj)], str(scratch))
    assert r.returncode == 0 and "DROPIN_OK" in r.stdout, r.stdout + r.stderr


def test_overlay_alone_reports_what_is_missing(tmp_path):
    """Without a project behind it the overlay still imports; a forwarded name fails with a message that says why."""
    code = textwrap.dedent("""
        from models.vae_gan import VaeGan
        from train.train_utils import PearsonCorrelation
        import train.train_utils as tu
        try:
            from train.train_utils import evaluate
        except ImportError:
            pass
        else:
            raise AssertionError('evaluate resolved without a project on the path')
        try:
            tu.evaluate
        except AttributeError as e:
            assert 'no other train/train_utils.py follows' in str(e), str(e)
            print('ALONE_OK')
        """)
    r = _run_child(code, [PKG], str(tmp_path))
    assert r.returncode == 0 and "ALONE_OK" in r.stdout, r.stdout + r.stderr
