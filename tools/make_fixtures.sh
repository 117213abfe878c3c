#!/bin/bash
# Regenerates everything under tests/golden from scratch (needs the reference tree at /root/reference).
#   1. synthetic clips (tools/gen_clip.py)
#   2. bitstreams with the reference encoder (oracle/_ref/TAppEncoder, cfg files of the reference, QP 32, MD5 SEI on)
#   3. syntax dumps + per-stage MD5s with the bridge decoder in dump mode (oracle/_ref/TAppDecoderBridge)
#   4. xz-compressed fixtures and hm_md5.json (the MD5s the reference decoder printed, all "(OK)")
# The 1080p encodes take 5-30 minutes each on one core; they run in parallel.
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cfg=/root/reference/cfg
work="$root/work"
mkdir -p "$work" && cd "$work"
make -s -j8 -C "$root/oracle" ref dump
python "$root/tools/gen_clip.py" 416 240 9 1234 c240.yuv
python "$root/tools/gen_clip.py" 1920 1080 60 20240607 c1080.yuv
python "$root/tools/gen_clip.py" 416 240 5 1234 n240.yuv noise
python "$root/tools/gen_clip.py" 416 240 9 1234 f240.yuv fade
python "$root/tools/gen_clip.py" 416 240 6 1234 n6.yuv noise
E="$root/oracle/_ref/TAppEncoder"
for c in lowdelay_P_main randomaccess_main randomaccess_he10 intra_main; do
  $E -c $cfg/encoder_$c.cfg -i c240.yuv -wdt 416 -hgt 240 -fr 30 -f 9 -b ${c}_240.bin -o ${c}_240_rec.yuv --SEIpictureDigest=1 > ${c}_240.enc.log &
done
$E -c $cfg/encoder_randomaccess_main.cfg -i c240.yuv -wdt 416 -hgt 240 -fr 30 -f 9 -b scalinglist_240.bin -o /dev/null --SEIpictureDigest=1 --ScalingList=1 > scalinglist_240.enc.log &
$E -c $cfg/encoder_lowdelay_P_main.cfg -i c240.yuv -wdt 416 -hgt 240 -fr 30 -f 6 -b lossless_240.bin -o /dev/null --SEIpictureDigest=1 --TransquantBypassEnableFlag=1 --CUTransquantBypassFlagValue=1 > lossless_240.enc.log &
# explicit weighted prediction on the fading clip: P slices (8-bit) and B slices (10-bit)
$E -c $cfg/encoder_lowdelay_P_main.cfg -i f240.yuv -wdt 416 -hgt 240 -fr 30 -f 9 -b wpp_240.bin -o /dev/null --SEIpictureDigest=1 -wpP 1 > wpp_240.enc.log &
$E -c $cfg/encoder_randomaccess_he10.cfg -i f240.yuv -wdt 416 -hgt 240 -fr 30 -f 9 -b wpb_240.bin -o /dev/null --SEIpictureDigest=1 -wpP 1 -wpB 1 > wpb_240.enc.log &
# constrained intra prediction; the noise patch yields intra blocks inside the B pictures
$E -c $cfg/encoder_randomaccess_main.cfg -i n6.yuv -wdt 416 -hgt 240 -fr 30 -f 6 -q 30 -b cip_240.bin -o /dev/null --SEIpictureDigest=1 --ConstrainedIntraPred=1 > cip_240.enc.log &
# 12-bit internal bit depth
$E -c $cfg/encoder_randomaccess_he10.cfg -i c240.yuv -wdt 416 -hgt 240 -fr 30 -f 5 -b bd12_240.bin -o /dev/null --SEIpictureDigest=1 --InternalBitDepth=12 > bd12_240.enc.log &
# six slices of five CTUs per picture
$E -c $cfg/encoder_randomaccess_main.cfg -i n240.yuv -wdt 416 -hgt 240 -fr 30 -f 5 -q 30 -b slices_240.bin -o /dev/null --SEIpictureDigest=1 --SliceMode=1 --SliceArgument=5 > slices_240.enc.log &
# 3 x 2 tiles
$E -c $cfg/encoder_randomaccess_main.cfg -i n240.yuv -wdt 416 -hgt 240 -fr 30 -f 5 -q 30 -b tiles_240.bin -o /dev/null --SEIpictureDigest=1 --UniformSpacingIdc=1 --NumTileColumnsMinus1=2 --NumTileRowsMinus1=1 > tiles_240.enc.log &
# the same two with in-loop filters that stop at slice / tile boundaries, and both at once on the plain clip
$E -c $cfg/encoder_randomaccess_main.cfg -i n240.yuv -wdt 416 -hgt 240 -fr 30 -f 5 -q 30 -b slices_nolf_240.bin -o /dev/null --SEIpictureDigest=1 --SliceMode=1 --SliceArgument=5 --LFCrossSliceBoundaryFlag=0 > slices_nolf_240.enc.log &
$E -c $cfg/encoder_randomaccess_main.cfg -i n240.yuv -wdt 416 -hgt 240 -fr 30 -f 5 -q 30 -b tiles_nolf_240.bin -o /dev/null --SEIpictureDigest=1 --UniformSpacingIdc=1 --NumTileColumnsMinus1=2 --NumTileRowsMinus1=1 --LFCrossTileBoundaryFlag=0 > tiles_nolf_240.enc.log &
$E -c $cfg/encoder_randomaccess_main.cfg -i c240.yuv -wdt 416 -hgt 240 -fr 30 -f 9 -q 36 -b slices_tiles_nolf_240.bin -o /dev/null --SEIpictureDigest=1 --SliceMode=1 --SliceArgument=3 --LFCrossSliceBoundaryFlag=0 --UniformSpacingIdc=1 --NumTileColumnsMinus1=1 --NumTileRowsMinus1=1 --LFCrossTileBoundaryFlag=0 > slices_tiles_nolf_240.enc.log &
# IPCM: noise patch at QP 6; 10-bit with 8-bit PCM samples and the in-loop filters off for IPCM units, 8-bit with them on
$E -c $cfg/encoder_randomaccess_he10.cfg -i n240.yuv -wdt 416 -hgt 240 -fr 30 -f 5 -q 6 -b pcm_240.bin -o /dev/null --SEIpictureDigest=1 --PCMEnabledFlag=1 --PCMFilterDisableFlag=1 > pcm_240.enc.log &
$E -c $cfg/encoder_lowdelay_P_main.cfg -i n240.yuv -wdt 416 -hgt 240 -fr 30 -f 4 -q 6 -b pcmlf_240.bin -o /dev/null --SEIpictureDigest=1 --PCMEnabledFlag=1 --PCMFilterDisableFlag=0 > pcmlf_240.enc.log &
$E -c $cfg/encoder_randomaccess_main.cfg -i c1080.yuv -wdt 1920 -hgt 1080 -fr 30 -f 60 -b ra_main_1080.bin -o /dev/null --SEIpictureDigest=1 > ra_main_1080.enc.log &
$E -c $cfg/encoder_randomaccess_he10.cfg -i c1080.yuv -wdt 1920 -hgt 1080 -fr 30 -f 60 -b ra_he10_1080.bin -o /dev/null --SEIpictureDigest=1 > ra_he10_1080.enc.log &
# all-intra: 16 frames as a syntax dump (4 MB), all 60 frames as a bitstream only (11 MB, kept beside the compiled
# reference under oracle/_ref/fixtures, not committed: decoded through the reference decoder's CLI with the engine behind it)
$E -c $cfg/encoder_intra_main.cfg -i c1080.yuv -wdt 1920 -hgt 1080 -fr 30 -f 16 -b intra_main_1080.bin -o /dev/null --SEIpictureDigest=1 > intra_main_1080.enc.log &
$E -c $cfg/encoder_intra_main.cfg -i c1080.yuv -wdt 1920 -hgt 1080 -fr 30 -f 60 -b intra_main_1080_f60.bin -o /dev/null --SEIpictureDigest=1 > intra_main_1080_f60.enc.log &
# the multi-stream benchmark's other seven 1080p random-access streams: clips of seeds 20240607 + k
for k in 1 2 3 4 5 6 7; do
  ( python "$root/tools/gen_clip.py" 1920 1080 60 $((20240607 + k)) c1080_s$k.yuv
    $E -c $cfg/encoder_randomaccess_main.cfg -i c1080_s$k.yuv -wdt 1920 -hgt 1080 -fr 30 -f 60 -b ra_main_1080_s$k.bin -o /dev/null --SEIpictureDigest=1 > ra_main_1080_s$k.enc.log
    rm -f c1080_s$k.yuv ) &
done
# option sweep: bitstreams only, decoded through the reference decoder's CLI with the engine behind it
sw() { name=$1; shift; $E "$@" -wdt 416 -hgt 240 -fr 30 -b sweep_$name.bin -o /dev/null --SEIpictureDigest=1 > sweep_$name.enc.log; }
sw dqp -c $cfg/encoder_randomaccess_main.cfg -i c240.yuv -f 5 --MaxDeltaQP=4 --MaxCuDQPDepth=2 --CbQpOffset=3 --CrQpOffset=-4 --DeblockingFilterControlPresent=1 --LoopFilterBetaOffset_div2=2 --LoopFilterTcOffset_div2=-1 &
sw q22 -c $cfg/encoder_lowdelay_P_main.cfg -i n240.yuv -f 4 -q 22 --SAOLcuBasedOptimization=0 &
sw q45 -c $cfg/encoder_randomaccess_he10.cfg -i c240.yuv -f 5 -q 45 --LoopFilterDisable=1 &
sw tu -c $cfg/encoder_randomaccess_main.cfg -i n240.yuv -f 5 -q 27 --QuadtreeTUMaxDepthIntra=3 --QuadtreeTUMaxDepthInter=3 --AMP=0 --SAO=0 --TransformSkip=0 &
sw lb -c $cfg/encoder_lowdelay_main.cfg -i f240.yuv -f 6 -q 30 &
wait
for n in dqp q22 q45 tu lb; do cp sweep_$n.bin "$root/tests/golden/"; done
D="$root/oracle/_ref/TAppDecoderBridge"
for s in lowdelay_P_main_240 randomaccess_main_240 randomaccess_he10_240 intra_main_240 scalinglist_240 lossless_240 pcm_240 pcmlf_240 wpp_240 wpb_240 cip_240 bd12_240 slices_240 tiles_240 slices_nolf_240 tiles_nolf_240 slices_tiles_nolf_240 ra_main_1080 ra_he10_1080 intra_main_1080 intra_main_1080_f60 ra_main_1080_s1 ra_main_1080_s2 ra_main_1080_s3 ra_main_1080_s4 ra_main_1080_s5 ra_main_1080_s6 ra_main_1080_s7; do
  HEVCRECON_MODE=dump HEVCRECON_DUMP=$s.syn $D -b $s.bin > $s.dec.log
done
# the reference decoder's output files at other bit depths than the internal one (digest per frame, display order)
R="$root/oracle/_ref/TAppDecoder"
$R -b randomaccess_he10_240.bin -o he10_d8.yuv -d 8 > /dev/null
$R -b randomaccess_he10_240.bin -o he10_d10.yuv -d 10 > /dev/null
$R -b lowdelay_P_main_240.bin -o ldp_d10.yuv -d 10 > /dev/null
$R -b bd12_240.bin -o bd12_d8.yuv -d 8 > /dev/null
cd "$root" && python - <<'PY'
import hashlib
out = {}
for name, f, ws in (("randomaccess_he10_240_d8", "work/he10_d8.yuv", 1), ("randomaccess_he10_240_d10", "work/he10_d10.yuv", 2),
                    ("lowdelay_P_main_240_d10", "work/ldp_d10.yuv", 2), ("bd12_240_d8", "work/bd12_d8.yuv", 1)):
    data = open(f, "rb").read()
    fb = 416 * 240 * 3 // 2 * ws
    out[name] = [hashlib.md5(data[i * fb:(i + 1) * fb]).hexdigest() for i in range(len(data) // fb)]
import json
json.dump(out, open("tests/golden/output_md5.json", "w"), indent=0)
PY
cd "$root" && python - <<'PY'
import json, re, shutil
from hevc_amd import syntax
names = {"lowdelay_P_main_240": "lowdelay_P_main_416x240", "randomaccess_main_240": "randomaccess_main_416x240",
         "randomaccess_he10_240": "randomaccess_he10_416x240", "intra_main_240": "intra_main_416x240",
         "scalinglist_240": "scalinglist_416x240", "lossless_240": "lossless_416x240",
         "pcm_240": "pcm_416x240", "pcmlf_240": "pcmlf_416x240", "wpp_240": "wpp_416x240", "wpb_240": "wpb_416x240", "cip_240": "cip_416x240", "bd12_240": "bd12_416x240", "slices_240": "slices_416x240", "tiles_240": "tiles_416x240",
         "slices_nolf_240": "slices_nolf_416x240", "tiles_nolf_240": "tiles_nolf_416x240", "slices_tiles_nolf_240": "slices_tiles_nolf_416x240",
         "ra_main_1080": "randomaccess_main_1920x1080", "ra_he10_1080": "randomaccess_he10_1920x1080",
         "intra_main_1080": "intra_main_1920x1080"}
names.update({"ra_main_1080_s%d" % k: "randomaccess_main_1920x1080_s%d" % k for k in range(1, 8)})
keys = {"ra_main_1080": "randomaccess_main_1080", "ra_he10_1080": "randomaccess_he10_1080"}
keys.update({"ra_main_1080_s%d" % k: "randomaccess_main_1080_s%d" % k for k in range(1, 8)})
md5 = {}
for src, dst in names.items():
    syntax.strip_planes_and_compress("work/%s.syn" % src, "tests/golden/%s.syn.xz" % dst, preset=6)
    rows = []
    for line in open("work/%s.dec.log" % src, errors="replace"):
        m = re.search(r"POC\s+(\d+).*\[MD5:([0-9a-f]{32}),([0-9a-f]{32}),([0-9a-f]{32}),\((\w+)\)\]", line)
        if m:
            assert m.group(5) == "OK"
            rows.append([int(m.group(1)), m.group(2), m.group(3), m.group(4)])
    md5[keys.get(src, src)] = rows
json.dump(md5, open("tests/golden/hm_md5.json", "w"), indent=0)
for b in ["lowdelay_P_main_240", "randomaccess_main_240", "randomaccess_he10_240", "intra_main_240", "scalinglist_240", "lossless_240", "pcm_240", "pcmlf_240", "wpp_240", "wpb_240", "cip_240", "bd12_240", "slices_240", "tiles_240", "slices_nolf_240", "tiles_nolf_240", "slices_tiles_nolf_240"]:
    shutil.copy("work/%s.bin" % b, "tests/golden/%s.bin" % b)
shutil.copy("work/ra_main_1080.bin", "tests/golden/randomaccess_main_1080.bin")
shutil.copy("work/ra_he10_1080.bin", "tests/golden/randomaccess_he10_1080.bin")
# the 60-frame all-intra stream: digests only (from its decoder log), bitstream beside the compiled reference
rows = []
for line in open("work/intra_main_1080_f60.dec.log", errors="replace"):
    m = re.search(r"POC\s+(\d+).*\[MD5:([0-9a-f]{32}),([0-9a-f]{32}),([0-9a-f]{32}),\((\w+)\)\]", line)
    if m:
        assert m.group(5) == "OK"
        rows.append([int(m.group(1)), m.group(2), m.group(3), m.group(4)])
md5 = json.load(open("tests/golden/hm_md5.json"))
md5["intra_main_1080_f60"] = rows
json.dump(md5, open("tests/golden/hm_md5.json", "w"), indent=0)
import os
os.makedirs("oracle/_ref/fixtures", exist_ok=True)
shutil.copy("work/intra_main_1080_f60.bin", "oracle/_ref/fixtures/intra_main_1080_f60.bin")
PY
