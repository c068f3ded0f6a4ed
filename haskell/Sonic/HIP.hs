-- | Sonic.HIP -- the reference's public surface (Sonic.SRS, Sonic.CommitmentScheme, Sonic.Protocol,
-- Sonic.Signature) over libsonic_hip.so, the MI355X prover path of this repository.
--
-- Every exported name has the type it has in sdiehl/sonic, except that `SRS` is an opaque handle to
-- device memory (its vectors are fetched on demand: `gNegativeX` .. `hPositiveAlphaX`, `srsPairing`)
-- instead of a record of lazily built vectors:
--
--   new          :: Int -> Fr -> Fr -> SRS                                       src/Sonic/SRS.hs:27-43
--   commitPoly   :: SRS -> Int -> VLaurent Fr -> G1 BLS12381                      src/Sonic/CommitmentScheme.hs:20-33
--   openPoly     :: SRS -> Fr -> VLaurent Fr -> (Fr, G1 BLS12381)                 src/Sonic/CommitmentScheme.hs:36-48
--   pcV          :: SRS -> Int -> G1 BLS12381 -> Fr -> (Fr, G1 BLS12381) -> Bool  src/Sonic/CommitmentScheme.hs:51-68
--   prove        :: MonadRandom m => SRS -> Assignment Fr -> ArithCircuit Fr -> m (Proof, RndOracle)
--                                                                                src/Sonic/Protocol.hs:47-109
--   verify       :: SRS -> ArithCircuit Fr -> Proof -> Fr -> Fr -> [(Fr, Fr)] -> Bool
--                                                                                src/Sonic/Protocol.hs:111-130
--   hscProve     :: MonadRandom m => SRS -> BiVLaurent Fr -> [(Fr, Fr)] -> m HscProof
--                                                                                src/Sonic/Signature.hs:32-72
--   hscVerify    :: SRS -> BiVLaurent Fr -> [(Fr, Fr)] -> HscProof -> Bool        src/Sonic/Signature.hs:74-90
--
-- plus what the reference has no counterpart for: a resident prover handle (`Prover`, `newProver`,
-- `prepare`, `proveWith`, `submit` / `collect`), one proof over several GPUs (`proveShared`), a list of
-- proofs over several GPUs (`proveBatch`), SRS persistence (`saveSRS`, `loadSRS`) and replication.
--
-- The prover's `rnd` draws stay in Haskell, in the reference's order (Protocol.hs:58,66,76,84-85;
-- Signature.hs:48,60); they cross the boundary as the explicit 8 + 2Q-element transcript, so
-- `RndOracle` is what it was.  Encodings: include/sonic_hip.h (Fr 32 bytes little-endian, G1 96 bytes
-- x || y little-endian, the point at infinity all zero, G2 192 bytes, GT 576 bytes).
--
-- Two one-token changes in the reference make the record constructors reachable from here:
--   src/Sonic/Protocol.hs:7    ( Proof      ->    ( Proof(..)
-- (HscProof(..) and RndOracle(..) are exported already).
--
-- NOT COMPILED in this repository: the build image has no ghc / cabal / stack.  tests/test_hs_shim.py
-- checks every `foreign import` below against include/sonic_hip.h (symbol exists, arity, and the kind
-- of every argument and of the result: pointer / Int64 / CInt / CSize), and tests/host/abi_harness.c
-- drives the same entry points from C99 with the calling convention `ccall` uses.
-- Dependency API used ([dep, unverified], as in SURVEY.md section 8c): galois-field 1.0.1 `fromP`, `toE`,
-- `fromE`, `toU'`; elliptic-curve 0.3.0 `Point(A, O)` of Data.Curve.Weierstrass; pairing 1.0.0 type names.
{-# LANGUAGE ForeignFunctionInterface #-}
{-# LANGUAGE RecordWildCards          #-}
{-# LANGUAGE EmptyDataDecls           #-}
module Sonic.HIP
  ( -- * Sonic.SRS
    SRS, new, newOn, replicate', srsD, srsDevice
  , gNegativeX, gPositiveX, gNegativeAlphaX, gPositiveAlphaX
  , hNegativeX, hPositiveX, hNegativeAlphaX, hPositiveAlphaX, srsPairing
  , saveSRS, loadSRS
    -- * Sonic.CommitmentScheme
  , commitPoly, openPoly, pcV
    -- * Sonic.Protocol
  , prove, proveWithTranscript, verify, decodeProof, encodeProof
    -- * Sonic.Signature
  , hscProve, hscVerify, decodeHscProof, encodeHscProof
    -- * resident handles, many GPUs
  , Prover, newProver, prepare, setAssignment, proveWith, submit, collect
  , proveShared, proveBatch, deviceCount
  ) where

import Protolude hiding (check)
import qualified Data.ByteString          as BS
import qualified Data.ByteString.Internal as BSI
import qualified Data.Vector              as V
import qualified GHC.Exts
import Control.Monad.Random (MonadRandom)
import Data.Curve.Weierstrass (Point(A, O))
import Data.Field.Galois (fromE, fromP, rnd, toE, toU')
import Data.Pairing.BLS12381 (BLS12381, Fq, Fq12, Fr, G1, G2, GT)
import Data.Poly.Sparse.Laurent (VLaurent)
import Bulletproofs.ArithmeticCircuit (ArithCircuit(..), Assignment(..), GateWeights(..))
import Foreign (FunPtr, ForeignPtr, Ptr, alloca, allocaArray, allocaBytes, castPtr, newForeignPtr, nullPtr,
                peek, pokeArray, withArrayLen, withForeignPtr)
import Foreign.C.String (CString, peekCString, withCString)
import Foreign.C.Types (CChar, CInt(..), CSize(..))
import System.IO.Unsafe (unsafePerformIO)

import Sonic.Protocol (Proof(..), RndOracle(..))
import Sonic.Signature (HscProof(..))
import Sonic.Utils (BiVLaurent)

-- ---------------------------------------------------------------------------------------------------------------------
-- the C ABI (include/sonic_hip.h); `safe`: the call may block for milliseconds and must not stop the RTS
-- ---------------------------------------------------------------------------------------------------------------------
data SrsHandle
data ProverHandle

foreign import ccall unsafe "sonic_abi_version"       c_abi_version      :: IO CInt
foreign import ccall safe   "sonic_init"              c_init             :: CInt -> IO CInt
foreign import ccall safe   "sonic_device_count"      c_device_count     :: Ptr CInt -> IO CInt
foreign import ccall unsafe "sonic_last_error"        c_last_error       :: Ptr CChar -> CSize -> IO CInt

foreign import ccall safe   "sonic_srs_new"           c_srs_new          :: Int64 -> Ptr Word8 -> Ptr Word8 -> Ptr (Ptr SrsHandle) -> IO CInt
foreign import ccall safe   "sonic_srs_new_on"        c_srs_new_on       :: CInt -> Int64 -> Ptr Word8 -> Ptr Word8 -> Ptr (Ptr SrsHandle) -> IO CInt
foreign import ccall safe   "sonic_srs_replicate"     c_srs_replicate    :: Ptr SrsHandle -> CInt -> Ptr (Ptr SrsHandle) -> IO CInt
foreign import ccall safe   "&sonic_srs_free"         p_srs_free         :: FunPtr (Ptr SrsHandle -> IO ())
foreign import ccall unsafe "sonic_srs_d"             c_srs_d            :: Ptr SrsHandle -> IO Int64
foreign import ccall unsafe "sonic_srs_device"        c_srs_device       :: Ptr SrsHandle -> IO CInt
foreign import ccall safe   "sonic_srs_get_points"    c_srs_get_points   :: Ptr SrsHandle -> CInt -> Int64 -> Int64 -> Ptr Word8 -> IO CInt
foreign import ccall safe   "sonic_srs_get_g2_points" c_srs_get_g2       :: Ptr SrsHandle -> CInt -> Int64 -> Int64 -> Ptr Word8 -> IO CInt
foreign import ccall safe   "sonic_srs_pairing"       c_srs_pairing      :: Ptr SrsHandle -> Ptr Word8 -> IO CInt
foreign import ccall safe   "sonic_srs_save"          c_srs_save         :: Ptr SrsHandle -> CString -> CInt -> IO CInt
foreign import ccall safe   "sonic_srs_load"          c_srs_load         :: CString -> Ptr (Ptr SrsHandle) -> IO CInt

foreign import ccall safe   "sonic_commit_poly"       c_commit_poly      :: Ptr SrsHandle -> Int64 -> Int64 -> Ptr Int64 -> Ptr Word8 -> Ptr Word8 -> IO CInt
foreign import ccall safe   "sonic_open_poly"         c_open_poly        :: Ptr SrsHandle -> Ptr Word8 -> Int64 -> Ptr Int64 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> IO CInt
foreign import ccall safe   "sonic_pc_v"              c_pc_v             :: Ptr SrsHandle -> Int64 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr CInt -> IO CInt

foreign import ccall unsafe "sonic_proof_size"        c_proof_size       :: Int64 -> IO CSize
foreign import ccall safe   "sonic_prove"             c_prove            :: Ptr SrsHandle -> Int64 -> Int64 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> IO CInt
foreign import ccall safe   "sonic_verify"            c_verify           :: Ptr SrsHandle -> Int64 -> Int64 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr CInt -> IO CInt

foreign import ccall unsafe "sonic_hsc_proof_size"    c_hsc_proof_size   :: Int64 -> IO CSize
foreign import ccall safe   "sonic_hsc_prove_poly"    c_hsc_prove_poly   :: Ptr SrsHandle -> Int64 -> Ptr Int64 -> Ptr Int64 -> Ptr Word8 -> Int64 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> IO CInt
foreign import ccall safe   "sonic_hsc_verify_poly"   c_hsc_verify_poly  :: Ptr SrsHandle -> Int64 -> Ptr Int64 -> Ptr Int64 -> Ptr Word8 -> Int64 -> Ptr Word8 -> Ptr Word8 -> Ptr CInt -> IO CInt

foreign import ccall safe   "sonic_prover_new"            c_prover_new     :: Ptr SrsHandle -> Int64 -> Int64 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr (Ptr ProverHandle) -> IO CInt
foreign import ccall safe   "&sonic_prover_free"          p_prover_free    :: FunPtr (Ptr ProverHandle -> IO ())
foreign import ccall safe   "sonic_prover_prepare"        c_prover_prepare :: Ptr ProverHandle -> IO CInt
foreign import ccall safe   "sonic_prover_set_assignment" c_prover_set     :: Ptr ProverHandle -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> IO CInt
foreign import ccall safe   "sonic_prover_prove"          c_prover_prove   :: Ptr ProverHandle -> Ptr Word8 -> Ptr Word8 -> IO CInt
foreign import ccall safe   "sonic_prover_submit"         c_prover_submit  :: Ptr ProverHandle -> Ptr Word8 -> IO CInt
foreign import ccall safe   "sonic_prover_collect"        c_prover_collect :: Ptr ProverHandle -> Ptr Word8 -> IO CInt
foreign import ccall safe   "sonic_prove_shared"          c_prove_shared   :: Ptr (Ptr ProverHandle) -> CInt -> Ptr Word8 -> Ptr Word8 -> IO CInt
foreign import ccall safe   "sonic_prove_batch"           c_prove_batch    :: Ptr (Ptr ProverHandle) -> CInt -> Int64 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr CInt -> IO CInt

-- ---------------------------------------------------------------------------------------------------------------------
-- status codes -> the reference's failure behaviour (it panics: Protocol.hs:55, CommitmentScheme.hs:70-73, :44)
-- ---------------------------------------------------------------------------------------------------------------------
lastError :: IO Text
lastError = allocaBytes 512 $ \buf -> do
  _ <- c_last_error buf 512
  toS <$> peekCString buf

check :: CInt -> IO ()
check 0 = pure ()
check 1 = lastError >>= panic                                     -- "Parameter d is not large enough: .."   Protocol.hs:55
check 2 = lastError >>= \m -> panic (m <> " (is not long enough)")  -- `index`                              CommitmentScheme.hs:70-73
check 4 = lastError >>= \m -> panic ("fromJust: " <> m)             -- inexact division                     CommitmentScheme.hs:44
check c = lastError >>= \m -> panic ("libsonic_hip status " <> show c <> ": " <> m)

abiExpected :: CInt
abiExpected = 6

-- every entry into the library goes through here once: the header this module was written against
libraryReady :: ()
libraryReady = unsafePerformIO $ do
  v <- c_abi_version
  when (v /= abiExpected) $ panic ("libsonic_hip: ABI version " <> show v <> ", this binding expects " <> show abiExpected)
  check =<< c_init (-1)
{-# NOINLINE libraryReady #-}

-- ---------------------------------------------------------------------------------------------------------------------
-- marshalling: little-endian integers, affine points, towers
-- ---------------------------------------------------------------------------------------------------------------------
leBytes :: Int -> Integer -> ByteString
leBytes len v = fst (BS.unfoldrN len (\x -> Just (fromIntegral (x `mod` 256), x `div` 256)) v)

leInteger :: ByteString -> Integer
leInteger = BS.foldr' (\b acc -> acc * 256 + fromIntegral b) 0

frToBytes :: Fr -> ByteString
frToBytes = leBytes 32 . fromP

frFromBytes :: ByteString -> Fr
frFromBytes = fromInteger . leInteger

fqToBytes :: Fq -> ByteString
fqToBytes = leBytes 48 . fromP

fqFromBytes :: ByteString -> Fq
fqFromBytes = fromInteger . leInteger

-- | `A x y` <-> x || y, `O` (mempty) <-> 96 zero bytes ((0, 0) is not on y^2 = x^3 + 4)
g1ToBytes :: G1 BLS12381 -> ByteString
g1ToBytes O       = BS.replicate 96 0
g1ToBytes (A x y) = fqToBytes x <> fqToBytes y

g1FromBytes :: ByteString -> G1 BLS12381
g1FromBytes b
  | BS.all (== 0) b = O
  | otherwise       = A (fqFromBytes (BS.take 48 b)) (fqFromBytes (BS.take 48 (BS.drop 48 b)))

-- | G2: x.c0 || x.c1 || y.c0 || y.c1, x = c0 + c1 u (Fq2 = Fq[u]/(u^2 + 1)); infinity = 192 zero bytes
g2FromBytes :: ByteString -> G2 BLS12381
g2FromBytes b
  | BS.all (== 0) b = O
  | otherwise       = A (toE [c 0, c 1]) (toE [c 2, c 3])
  where c i = fqFromBytes (BS.take 48 (BS.drop (48 * i) b))

-- | GT: the Fq12 element as its twelve Fq coefficients in tower order, c[i][j][k] at 48 (6 i + 2 j + k)
--   (Fq12 = Fq6[w]/(w^2 - v), Fq6 = Fq2[v]/(v^3 - (1 + u)), Fq2 = Fq[u]/(u^2 + 1))
fq12FromBytes :: ByteString -> Fq12
fq12FromBytes b = toE [ toE [ toE [c i j 0, c i j 1] | j <- [0, 1, 2] ] | i <- [0, 1] ]
  where c i j k = fqFromBytes (BS.take 48 (BS.drop (48 * (6 * i + 2 * j + k)) b))

fq12ToBytes :: Fq12 -> ByteString
fq12ToBytes x = BS.concat [ fqToBytes c | h <- fromE x, f2 <- fromE h, c <- fromE f2 ]

withBytes :: ByteString -> (Ptr Word8 -> IO a) -> IO a
withBytes b f = BS.useAsCString b (f . castPtr)             -- (copies; never a null pointer, also for an empty list)

withFr :: Fr -> (Ptr Word8 -> IO a) -> IO a
withFr = withBytes . frToBytes

withFrs :: [Fr] -> (Ptr Word8 -> IO a) -> IO a
withFrs = withBytes . BS.concat . map frToBytes

withG1 :: G1 BLS12381 -> (Ptr Word8 -> IO a) -> IO a
withG1 = withBytes . g1ToBytes

-- | the `(Int, Fr)` list that `GHC.Exts.toList` yields for a VLaurent (CommitmentScheme.hs:33,48)
withTerms :: [(Int, Fr)] -> (Int64 -> Ptr Int64 -> Ptr Word8 -> IO a) -> IO a
withTerms ts f =
  withArrayLen (map (fromIntegral . fst) ts) $ \n es ->
    withFrs (map snd ts) $ \cs -> f (fromIntegral n) es cs

-- | BiVLaurent Fr = VLaurent (VLaurent Fr): X outside, Y inside (Utils.hs:15-21) -> three parallel arrays
withBiTerms :: BiVLaurent Fr -> (Int64 -> Ptr Int64 -> Ptr Int64 -> Ptr Word8 -> IO a) -> IO a
withBiTerms sXY f =
  let ts = [ (i, j, c) | (i, inner) <- GHC.Exts.toList sXY, (j, c) <- GHC.Exts.toList inner ]
  in withArrayLen [ fromIntegral i | (i, _, _) <- ts ] $ \n xs ->
       withArrayLen [ fromIntegral j | (_, j, _) <- ts ] $ \_ ys ->
         withFrs [ c | (_, _, c) <- ts ] $ \cs -> f (fromIntegral n) xs ys cs

packed :: Ptr Word8 -> Int -> IO ByteString
packed p len = BS.packCStringLen (castPtr p, len)

chunks :: Int -> ByteString -> [ByteString]
chunks k b | BS.null b = []
           | otherwise = BS.take k b : chunks k (BS.drop k b)

-- ---------------------------------------------------------------------------------------------------------------------
-- Sonic.SRS
-- ---------------------------------------------------------------------------------------------------------------------
-- | the reference's record of ten fields (SRS.hs:11-22) as a handle: d and a finalizer; the vectors live in HBM
newtype SRS = SRS (ForeignPtr SrsHandle)

wrapSRS :: Ptr (Ptr SrsHandle) -> IO SRS
wrapSRS out = SRS <$> (newForeignPtr p_srs_free =<< peek out)

-- | SRS.new :: Int -> Fr -> Fr -> SRS   (src/Sonic/SRS.hs:27-43); generated on the default GPU
new :: Int -> Fr -> Fr -> SRS
new d x alpha = libraryReady `seq` unsafePerformIO (
  withFr x $ \px -> withFr alpha $ \pa -> alloca $ \out -> do
    check =<< c_srs_new (fromIntegral d) px pa out
    wrapSRS out)
{-# NOINLINE new #-}

-- | the same on a named GPU
newOn :: Int -> Int -> Fr -> Fr -> IO SRS
newOn device d x alpha = libraryReady `seq`
  withFr x $ \px -> withFr alpha $ \pa -> alloca $ \out -> do
    check =<< c_srs_new_on (fromIntegral device) (fromIntegral d) px pa out
    wrapSRS out

-- | a replica on another GPU (device-to-device copy of the bases and their window tables)
replicate' :: SRS -> Int -> IO SRS
replicate' (SRS h) device = withForeignPtr h $ \p -> alloca $ \out -> do
  check =<< c_srs_replicate p (fromIntegral device) out
  wrapSRS out

srsD :: SRS -> Int
srsD (SRS h) = unsafePerformIO (withForeignPtr h (fmap fromIntegral . c_srs_d))

srsDevice :: SRS -> Int
srsDevice (SRS h) = unsafePerformIO (withForeignPtr h (fmap fromIntegral . c_srs_device))

-- exponents e0 .. e0 + n - 1 of one basis (0: g^{x^e}, 1: g^{alpha x^e}); SRS.hs:33-39 index maps in include/sonic_hip.h
g1Range :: SRS -> CInt -> Int -> Int -> [G1 BLS12381]
g1Range (SRS h) basis e0 n = unsafePerformIO $
  withForeignPtr h $ \p -> allocaBytes (96 * max 1 n) $ \out -> do
    check =<< c_srs_get_points p basis (fromIntegral e0) (fromIntegral n) out
    map g1FromBytes . chunks 96 <$> packed out (96 * n)

g2Range :: SRS -> CInt -> Int -> Int -> [G2 BLS12381]
g2Range (SRS h) basis e0 n = unsafePerformIO $
  withForeignPtr h $ \p -> allocaBytes (192 * max 1 n) $ \out -> do
    check =<< c_srs_get_g2 p basis (fromIntegral e0) (fromIntegral n) out
    map g2FromBytes . chunks 192 <$> packed out (192 * n)

gNegativeX, gPositiveX, gNegativeAlphaX, gPositiveAlphaX :: SRS -> V.Vector (G1 BLS12381)
gNegativeX      s = V.fromList (reverse (g1Range s 0 (negate (srsD s)) (srsD s)))       -- [k] = g^{x^{-(k+1)}}
gPositiveX      s = V.fromList (g1Range s 0 0 (srsD s + 1))                             -- [k] = g^{x^k}
gNegativeAlphaX s = V.fromList (reverse (g1Range s 1 (negate (srsD s)) (srsD s)))       -- [k] = g^{alpha x^{-(k+1)}}
gPositiveAlphaX s = V.fromList (g1Range s 1 1 (srsD s))                                 -- [k] = g^{alpha x^{k+1}}: g^alpha is not shared

hNegativeX, hPositiveX, hNegativeAlphaX, hPositiveAlphaX :: SRS -> V.Vector (G2 BLS12381)
hNegativeX      s = V.fromList (reverse (g2Range s 0 (negate (srsD s)) (srsD s)))
hPositiveX      s = V.fromList (g2Range s 0 0 (srsD s + 1))
hNegativeAlphaX s = V.fromList (reverse (g2Range s 1 (negate (srsD s)) (srsD s)))
hPositiveAlphaX s = V.fromList (g2Range s 1 0 (srsD s + 1))                             -- [0] = h^alpha (SRS.hs:41)

-- | srsPairing = pairing gen (hPositiveAlphaX !! 0)    (src/Sonic/SRS.hs:21,42)
srsPairing :: SRS -> GT BLS12381
srsPairing (SRS h) = unsafePerformIO $
  withForeignPtr h $ \p -> allocaBytes 576 $ \out -> do
    check =<< c_srs_pairing p out
    toU' . fq12FromBytes <$> packed out 576

-- | persistence (nothing in the reference): withG2 = write the verifier half as well
saveSRS :: SRS -> FilePath -> Bool -> IO ()
saveSRS (SRS h) path withG2 = withForeignPtr h $ \p -> withCString path $ \cp ->
  check =<< c_srs_save p cp (if withG2 then 1 else 0)

loadSRS :: FilePath -> IO SRS
loadSRS path = libraryReady `seq` withCString path (\cp -> alloca $ \out -> do
  check =<< c_srs_load cp out
  wrapSRS out)

-- ---------------------------------------------------------------------------------------------------------------------
-- Sonic.CommitmentScheme
-- ---------------------------------------------------------------------------------------------------------------------
-- | commitPoly :: SRS -> Int -> VLaurent Fr -> G1 BLS12381      (src/Sonic/CommitmentScheme.hs:20-33)
commitPoly :: SRS -> Int -> VLaurent Fr -> G1 BLS12381
commitPoly (SRS h) maxm fX = unsafePerformIO $
  withForeignPtr h $ \p -> withTerms (GHC.Exts.toList fX) $ \n es cs ->
    allocaBytes 96 $ \out -> do
      check =<< c_commit_poly p (fromIntegral maxm) n es cs out
      g1FromBytes <$> packed out 96

-- | openPoly :: SRS -> Fr -> VLaurent Fr -> (Fr, G1 BLS12381)   (src/Sonic/CommitmentScheme.hs:36-48)
openPoly :: SRS -> Fr -> VLaurent Fr -> (Fr, G1 BLS12381)
openPoly (SRS h) z fX = unsafePerformIO $
  withForeignPtr h $ \p -> withFr z $ \pz -> withTerms (GHC.Exts.toList fX) $ \n es cs ->
    allocaBytes 32 $ \fz -> allocaBytes 96 $ \out -> do
      check =<< c_open_poly p pz n es cs fz out
      (,) <$> (frFromBytes <$> packed fz 32) <*> (g1FromBytes <$> packed out 96)

-- | pcV :: SRS -> Int -> G1 BLS12381 -> Fr -> (Fr, G1 BLS12381) -> Bool   (src/Sonic/CommitmentScheme.hs:51-68)
pcV :: SRS -> Int -> G1 BLS12381 -> Fr -> (Fr, G1 BLS12381) -> Bool
pcV (SRS h) maxm commitment z (v, w) = unsafePerformIO $
  withForeignPtr h $ \p -> withG1 commitment $ \pc -> withFr z $ \pz -> withFr v $ \pv -> withG1 w $ \pw ->
    alloca $ \acc -> do
      check =<< c_pc_v p (fromIntegral maxm) pc pz pv pw acc
      (/= 0) <$> peek acc

-- ---------------------------------------------------------------------------------------------------------------------
-- proof bytes <-> the records (include/sonic_hip.h: record order of `Proof`, Protocol.hs:28-38, then `HscProof`,
-- Signature.hs:22-29):  R T a Wa b Wb Wt s  [S_j s_j W_j]_j  [s'_j W'_j Q_j]_j  Qv C u v
-- ---------------------------------------------------------------------------------------------------------------------
takeG :: ByteString -> (G1 BLS12381, ByteString)
takeG b = (g1FromBytes (BS.take 96 b), BS.drop 96 b)

takeF :: ByteString -> (Fr, ByteString)
takeF b = (frFromBytes (BS.take 32 b), BS.drop 32 b)

-- | the HscProof part: (2 + 4 m) G1 + (2 + 2 m) Fr
decodeHscProof :: Int -> ByteString -> HscProof
decodeHscProof m b0 =
  let goS :: Int -> ByteString -> ([(G1 BLS12381, (Fr, G1 BLS12381))], ByteString)
      goS 0 b = ([], b)
      goS k b = let (sj, b1) = takeG b; (ev, b2) = takeF b1; (wj, b3) = takeG b2
                    (rest, b4) = goS (k - 1) b3
                in ((sj, (ev, wj)) : rest, b4)
      goW :: Int -> ByteString -> ([(Fr, G1 BLS12381, G1 BLS12381)], ByteString)
      goW 0 b = ([], b)
      goW k b = let (ev, b1) = takeF b; (wj', b2) = takeG b1; (qj, b3) = takeG b2
                    (rest, b4) = goW (k - 1) b3
                in ((ev, wj', qj) : rest, b4)
      (ss, c1) = goS m b0
      (ws, c2) = goW m c1
      (qv, c3) = takeG c2
      (cc, c4) = takeG c3
      (u, c5)  = takeF c4
      (v, _)   = takeF c5
  in HscProof { hscS = ss, hscW = ws, hscQv = qv, hscC = cc, hscU = u, hscV = v }

encodeHscProof :: HscProof -> ByteString
encodeHscProof HscProof{..} = BS.concat $
  [ g1ToBytes sj <> frToBytes ev <> g1ToBytes wj | (sj, (ev, wj)) <- hscS ] ++
  [ frToBytes ev <> g1ToBytes wj' <> g1ToBytes qj | (ev, wj', qj) <- hscW ] ++
  [ g1ToBytes hscQv, g1ToBytes hscC, frToBytes hscU, frToBytes hscV ]

-- | (7 + 4Q) G1 + (5 + 2Q) Fr
decodeProof :: Int -> ByteString -> Proof
decodeProof q b0 =
  let (r, b1)  = takeG b0
      (t, b2)  = takeG b1
      (a, b3)  = takeF b2
      (wa, b4) = takeG b3
      (b, b5)  = takeF b4
      (wb, b6) = takeG b5
      (wt, b7) = takeG b6
      (s, b8)  = takeF b7
  in Proof { prR = r, prT = t, prA = a, prWa = wa, prB = b, prWb = wb, prWt = wt, prS = s
           , prHscProof = decodeHscProof q b8 }

encodeProof :: Proof -> ByteString
encodeProof Proof{..} = BS.concat
  [ g1ToBytes prR, g1ToBytes prT, frToBytes prA, g1ToBytes prWa, frToBytes prB, g1ToBytes prWb, g1ToBytes prWt, frToBytes prS
  , encodeHscProof prHscProof ]

-- ---------------------------------------------------------------------------------------------------------------------
-- Sonic.Protocol
-- ---------------------------------------------------------------------------------------------------------------------
-- dense Q x n row-major weights, as include/sonic_hip.h takes them
withCircuit :: ArithCircuit Fr -> (Int64 -> Int64 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> IO a) -> IO a
withCircuit ArithCircuit{..} f =
  let GateWeights{..} = weights
      q = length wL
      n = case wL of { (row : _) -> length row; [] -> 0 }
  in withFrs (concat wL) $ \pwl -> withFrs (concat wR) $ \pwr -> withFrs (concat wO) $ \pwo -> withFrs cs $ \pcs ->
       f (fromIntegral n) (fromIntegral q) pwl pwr pwo pcs

-- | the draws of `prove` + `hscProve` in the reference's order: c_{n+1..n+4}, y, z, y_1..y_Q, z_1..z_Q, u, v
drawTranscript :: MonadRandom m => Int -> m ([Fr], RndOracle)
drawTranscript q = do
  cns <- replicateM 4 rnd               -- Protocol.hs:58
  y   <- rnd                            -- :66
  z   <- rnd                            -- :76
  ys  <- replicateM q rnd               -- :84
  zs  <- replicateM q rnd               -- :85
  u   <- rnd                            -- Signature.hs:48
  v   <- rnd                            -- :60
  pure (cns ++ [y, z] ++ ys ++ zs ++ [u, v], RndOracle { rndOracleY = y, rndOracleZ = z, rndOracleYZs = zip ys zs })

-- | prove with the draws supplied by the caller (what makes proofs reproducible; the tests of this repository use it)
proveWithTranscript :: SRS -> Assignment Fr -> ArithCircuit Fr -> [Fr] -> Proof
proveWithTranscript (SRS h) Assignment{..} circuit transcript = unsafePerformIO $
  withForeignPtr h $ \p -> withCircuit circuit $ \_ q pwl pwr pwo pcs ->
    withFrs aL $ \pal -> withFrs aR $ \par -> withFrs aO $ \pao -> withFrs transcript $ \ptr -> do
      sz <- fromIntegral <$> c_proof_size q
      bytes <- BSI.create sz $ \out ->
        check =<< c_prove p (fromIntegral (length aL)) q pwl pwr pwo pcs pal par pao ptr out
      pure (decodeProof (fromIntegral q) bytes)

-- | prove :: MonadRandom m => SRS -> Assignment Fr -> ArithCircuit Fr -> m (Proof, RndOracle)
--   (src/Sonic/Protocol.hs:47-109, with hscProve, src/Sonic/Signature.hs:38-72, inside)
prove :: MonadRandom m => SRS -> Assignment Fr -> ArithCircuit Fr -> m (Proof, RndOracle)
prove srs assignment circuit = do
  (transcript, oracle) <- drawTranscript (length (wL (weights circuit)))
  pure (proveWithTranscript srs assignment circuit transcript, oracle)

-- | verify :: SRS -> ArithCircuit Fr -> Proof -> Fr -> Fr -> [(Fr, Fr)] -> Bool   (src/Sonic/Protocol.hs:111-130)
--   host CPU inside the library (its own pairing); the reference's `verify` on the points `decodeProof` returns is equivalent
verify :: SRS -> ArithCircuit Fr -> Proof -> Fr -> Fr -> [(Fr, Fr)] -> Bool
verify (SRS h) circuit proof y z yzs = unsafePerformIO $
  withForeignPtr h $ \p -> withCircuit circuit $ \n q pwl pwr pwo pcs ->
    withBytes (encodeProof proof) $ \ppf -> withFr y $ \py -> withFr z $ \pz ->
      withFrs (concat [ [yj, zj] | (yj, zj) <- yzs ]) $ \pyz -> alloca $ \acc -> do
        check =<< c_verify p n q pwl pwr pwo pcs ppf py pz pyz acc
        (/= 0) <$> peek acc

-- ---------------------------------------------------------------------------------------------------------------------
-- Sonic.Signature
-- ---------------------------------------------------------------------------------------------------------------------
-- | hscProve :: MonadRandom m => SRS -> BiVLaurent Fr -> [(Fr, Fr)] -> m HscProof   (src/Sonic/Signature.hs:32-72)
hscProve :: MonadRandom m => SRS -> BiVLaurent Fr -> [(Fr, Fr)] -> m HscProof
hscProve (SRS h) sXY yzs = do
  u <- rnd                              -- Signature.hs:48
  v <- rnd                              -- :60
  let m = length yzs
  pure $ unsafePerformIO $
    withForeignPtr h $ \p -> withBiTerms sXY $ \nt xs ys cs ->
      withFrs (concat [ [yj, zj] | (yj, zj) <- yzs ]) $ \pyz -> withFr u $ \pu -> withFr v $ \pv -> do
        sz <- fromIntegral <$> c_hsc_proof_size (fromIntegral m)
        bytes <- BSI.create sz $ \out ->
          check =<< c_hsc_prove_poly p nt xs ys cs (fromIntegral m) pyz pu pv out
        pure (decodeHscProof m bytes)

-- | hscVerify :: SRS -> BiVLaurent Fr -> [(Fr, Fr)] -> HscProof -> Bool   (src/Sonic/Signature.hs:74-90)
hscVerify :: SRS -> BiVLaurent Fr -> [(Fr, Fr)] -> HscProof -> Bool
hscVerify (SRS h) sXY yzs proof = unsafePerformIO $
  withForeignPtr h $ \p -> withBiTerms sXY $ \nt xs ys cs ->
    withFrs (concat [ [yj, zj] | (yj, zj) <- yzs ]) $ \pyz -> withBytes (encodeHscProof proof) $ \ph ->
      alloca $ \acc -> do
        check =<< c_hsc_verify_poly p nt xs ys cs (fromIntegral (length yzs)) pyz ph acc
        (/= 0) <$> peek acc

-- ---------------------------------------------------------------------------------------------------------------------
-- resident handles: circuit (and assignment) stay in HBM between proofs -- `mapM (prove srs asg) circuits` without re-uploading
-- ---------------------------------------------------------------------------------------------------------------------
data Prover = Prover { proverHandle :: ForeignPtr ProverHandle, proverSrs :: SRS, proverQ :: Int }

newProver :: SRS -> ArithCircuit Fr -> IO Prover
newProver srs@(SRS h) circuit =
  withForeignPtr h $ \p -> withCircuit circuit $ \n q pwl pwr pwo pcs -> alloca $ \out -> do
    check =<< c_prover_new p n q pwl pwr pwo pcs out
    fp <- newForeignPtr p_prover_free =<< peek out
    pure Prover { proverHandle = fp, proverSrs = srs, proverQ = fromIntegral q }       -- (keeps the SRS alive as long as the handle)

-- | once per circuit: commits the Q constraint rows of sPoly (Constraints.hs:34-53); same proof bytes afterwards, fewer terms
prepare :: Prover -> IO ()
prepare Prover{..} = withForeignPtr proverHandle (check <=< c_prover_prepare)

setAssignment :: Prover -> Assignment Fr -> IO ()
setAssignment Prover{..} Assignment{..} =
  withForeignPtr proverHandle $ \p -> withFrs aL $ \pal -> withFrs aR $ \par -> withFrs aO $ \pao ->
    check =<< c_prover_set p pal par pao

proofBytes :: Int -> (Ptr Word8 -> IO CInt) -> IO ByteString
proofBytes q body = do
  sz <- fromIntegral <$> c_proof_size (fromIntegral q)
  BSI.create sz (check <=< body)

-- | prove on a resident handle (Protocol.hs:47-109); the draws are made here exactly as `prove` makes them
proveWith :: Prover -> IO (Proof, RndOracle)
proveWith Prover{..} = do
  (transcript, oracle) <- drawTranscript proverQ
  bytes <- withForeignPtr proverHandle $ \p -> withFrs transcript $ \ptr ->
    proofBytes proverQ (c_prover_prove p ptr)
  pure (decodeProof proverQ bytes, oracle)

-- | the two halves of `proveWith`: `submit` queues the proof on the GPU and returns, `collect` waits and finishes it.  A thread that
--   alternates between two handles (submit a; submit b; collect a; submit a; collect b; ..) streams a list of proofs.
submit :: Prover -> IO RndOracle
submit Prover{..} = do
  (transcript, oracle) <- drawTranscript proverQ
  withForeignPtr proverHandle $ \p -> withFrs transcript $ \ptr -> check =<< c_prover_submit p ptr
  pure oracle

collect :: Prover -> IO Proof
collect Prover{..} =
  decodeProof proverQ <$> withForeignPtr proverHandle (\p -> proofBytes proverQ (c_prover_collect p))

withProvers :: [Prover] -> (Ptr (Ptr ProverHandle) -> CInt -> IO a) -> IO a
withProvers ps f = go ps []
  where
    go [] acc = let hs = reverse acc in allocaArray (length hs) $ \arr -> pokeArray arr hs >> f arr (fromIntegral (length hs))
    go (Prover{..} : rest) acc = withForeignPtr proverHandle $ \p -> go rest (p : acc)

-- | ONE proof made by several GPUs: handle r (on its own GPU, over a replica of the SRS, same circuit and assignment) runs rank r's
--   share of the proof's 7 + 4Q MSMs; same bytes as one GPU's proof
proveShared :: [Prover] -> IO (Proof, RndOracle)
proveShared [] = panic "proveShared: no handles"
proveShared ps@(p0 : _) = do
  let q = proverQ p0
  (transcript, oracle) <- drawTranscript q
  bytes <- withProvers ps $ \arr world -> withFrs transcript $ \ptr ->
    proofBytes q (c_prove_shared arr world ptr)
  pure (decodeProof q bytes, oracle)

-- | `mapM (\asg -> prove srs asg circuit) assignments` over the handles' GPUs (proof i on handle i mod length handles; two handles per
--   GPU stream that GPU's proofs); no collective
proveBatch :: [Prover] -> [Assignment Fr] -> IO [(Proof, RndOracle)]
proveBatch [] _ = panic "proveBatch: no handles"
proveBatch ps@(p0 : _) asgs = do
  let q = proverQ p0
      k = length asgs
  drawn <- replicateM k (drawTranscript q)
  psz <- fromIntegral <$> c_proof_size (fromIntegral q)
  bytes <- withProvers ps $ \arr np ->
    withFrs (concatMap aL asgs) $ \pal -> withFrs (concatMap aR asgs) $ \par -> withFrs (concatMap aO asgs) $ \pao ->
      withFrs (concatMap fst drawn) $ \ptr ->
        BSI.create (psz * k) $ \out ->
          check =<< c_prove_batch arr np (fromIntegral k) pal par pao ptr out nullPtr
  pure (zip (map (decodeProof q) (chunks psz bytes)) (map snd drawn))

deviceCount :: IO Int
deviceCount = alloca $ \out -> do
  _ <- c_device_count out                -- (SONIC_ERR_NO_DEVICE and 0 without a GPU: there is no CPU fallback)
  fromIntegral <$> peek out
