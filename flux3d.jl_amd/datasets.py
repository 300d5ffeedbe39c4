"""The on-disk side of the evaluation path: ModelNet10 / ModelNet40 listings over a directory of OFF
files, the way src/datasets/modelnet/base.jl:30-108 walks them -- ``<root>/ModelNet<variant>/<category>/<split>/*.off``,
one (category, path) per file in category order, ``dset[i]`` = ``load_trimesh(path)`` (base.jl:100-101).  Host-side
file handling only: nothing here touches the GPU, and nothing is downloaded (the reference's ``download=true`` fetches
the Princeton archive; this environment has no network -- ``extract`` unpacks an archive that is already on disk, the
``unzip`` of base.jl:87).  Transforms other than ``sample_points`` are out of scope."""
import os
import zipfile

from .rep import load_trimesh

MN10_CLASSES = ["bathtub", "bed", "chair", "desk", "dresser", "monitor", "night_stand", "sofa", "table", "toilet"]
MN40_CLASSES = ["airplane", "bathtub", "bed", "bench", "bookshelf", "bottle", "bowl", "car", "chair", "cone", "cup",
                "curtain", "desk", "door", "dresser", "flower_pot", "glass_box", "guitar", "keyboard", "lamp", "laptop",
                "mantel", "monitor", "night_stand", "person", "piano", "plant", "radio", "range_hood", "sink", "sofa",
                "stairs", "stool", "table", "tent", "toilet", "tv_stand", "vase", "wardrobe", "xbox"]


def extract(root, variant):
    """``<root>/ModelNet<variant>.zip`` -> ``<root>/ModelNet<variant>/`` unless it is there already (base.jl:70-97
    without the download); returns the directory."""
    local_dir = os.path.join(root, f"ModelNet{variant}")
    if not os.path.isdir(local_dir):
        local_zip = os.path.join(root, f"ModelNet{variant}.zip")
        if not os.path.isfile(local_zip):
            raise FileNotFoundError("dataset not found and auto-download is not available here")
        with zipfile.ZipFile(local_zip) as z:
            z.extractall(root)
    return local_dir


class ModelNet:
    """``ModelNet(root, variant=10, train=True, categories=None)``: ``len(d)``, ``d.datapaths`` = [(category, path)],
    ``d[i]`` = (TriMesh, class index (1-based, as the reference's ``classes_to_idx``), category)."""

    def __init__(self, root, variant=10, train=True, categories=None):
        if variant not in (10, 40):
            raise ValueError("ModelNet variant must be 10 or 40")
        valid = MN10_CLASSES if variant == 10 else MN40_CLASSES
        self.root = os.path.normpath(root)
        self.path = extract(self.root, variant)
        self.train = bool(train)
        self.categories = list(categories) if categories is not None else list(valid)
        split = "train" if train else "test"
        self.datapaths = []
        for category in self.categories:
            if category not in valid:
                raise ValueError(f"given category: {category} is not a valid ModelNet{variant} category.")
            d = os.path.join(self.path, category, split)
            self.datapaths += [(category, os.path.join(d, f)) for f in sorted(os.listdir(d)) if f.split(".")[-1] == "off"]
        self.classes_to_idx = {c: i + 1 for i, c in enumerate(self.categories)}

    def __len__(self):
        return len(self.datapaths)

    def __getitem__(self, idx):
        category, path = self.datapaths[idx]
        return load_trimesh(path), self.classes_to_idx[category], category


def ModelNet10(root, train=True, categories=None):
    return ModelNet(root, 10, train, categories)


def ModelNet40(root, train=True, categories=None):
    return ModelNet(root, 40, train, categories)
