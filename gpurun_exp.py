import os, sys, time, numpy as np
sys.path.insert(0,'mpeg-pcc-tmc2_amd')
import tmc2_amd as T
ctx=T.Context(0)
clouds=[T.synth_cloud('longdress_vox10',f) for f in range(3)]
for steps in ('0','1','2','3'):
    os.environ['TMC2_REFINE_CLOSURE_STEPS']=steps
    rep=0; tt=0
    for xyz,rgb in clouds:
        fr=ctx.frame(xyz,rgb); fr.normals_compute(16,1); w=fr.weight_normal(11,0.6); fr.segmenter_initial_segmentation(w)
        ctx.stage_reset(); t=time.time(); fr.segmenter_refine_grid_based(1024,3.0,50,4,192); ctx.synchronize(); tt+=time.time()-t
        rep+=ctx.stage_calls().get('refine_closure_replays',0)
    print('closure steps',steps,'replays',rep,'of',len(clouds),'refine wall %.1f ms/frame'%(1000*tt/len(clouds)))
os.environ['TMC2_REFINE_CLOSURE_STEPS']='3'
xyz,rgb=clouds[0]
for rep in range(2):
    fr=ctx.frame(xyz,rgb); ctx.stage_reset(); t=time.time()
    w=fr.weight_normal(11,0.6); p=T.ctc_params(50,11,w); fr.segmenter_compute(p); h=fr.encoder_pack_flexible(1280,2,1.0)
    W,H=T.encoder_canvas_size([h]); fr.encoder_generate_geometry_images(W,H,4); fr.encoder_generate_attribute_images(); ctx.synchronize()
    print('single frame wall %.1f ms'%(1000*(time.time()-t)), {k:round(v,2) for k,v in ctx.stage_ms().items()}, {k:v for k,v in ctx.stage_calls().items() if k.startswith('k:')})
