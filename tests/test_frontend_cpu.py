"""host logic of the batch-dict entry (utils/custom_dataset_sdxl.py:30, train_sdxl_zh.py:386-389)"""
import pytest
import torch


def test_add_time_ids_and_buckets():
    from pea_diffusion_amd._lib import PeaError
    from pea_diffusion_amd.frontend import BUCKETS, add_time_ids_from_batch
    assert BUCKETS[4] == [640, 640] and len(BUCKETS) == 9 and BUCKETS[0] == [448, 896] and BUCKETS[8] == [896, 448]
    b = {"original_size": torch.tensor([[1000, 800], [640, 700]]), "crops_coords_top_left": torch.tensor([[3, 5], [0, 9]]),
         "bucket_id": torch.tensor(2)}
    t = add_time_ids_from_batch(b, "cpu")
    assert t.dtype == torch.float32 and t.tolist() == [[1000, 800, 3, 5, 512, 768], [640, 700, 0, 9, 512, 768]]
    with pytest.raises(PeaError):
        add_time_ids_from_batch(dict(b, bucket_id=torch.tensor(9)), "cpu")
